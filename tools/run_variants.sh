for v in sdwa single; do
  echo "== $v"
  MALCS_LIB=build_tools/libmalcs_$v.so timeout -k 10 100 python tools/time_window.py 100000 1000000 || exit 1
done
MALCS_LIB=build_tools/libmalcs_sdwa.so timeout -k 10 100 python tests/checks/fuzz_window.py 60000 11 4 || exit 1
