for v in base la4 la8 al4 base; do
  echo "== $v"
  MALCS_LIB=build_tools/libmalcs_$v.so timeout -k 10 100 python tools/time_window.py 100000 100000 1000000 || exit 1
done
