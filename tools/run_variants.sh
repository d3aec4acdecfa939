for v in nt nont nt nont; do
  echo "== $v"
  MALCS_LIB=build_tools/libmalcs_$v.so timeout -k 10 100 python tools/time_window.py 100000 100000 1000000 || exit 1
done
MALCS_LIB=build_tools/libmalcs_nt.so timeout -k 10 100 python tools/diag_timeline.py 100000 | head -8
