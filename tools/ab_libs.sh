python bench.py --steps 8 --no-cpu-baseline >/dev/null 2>&1
for lib in ${LIBS:-libllamahip.so libllamahip_b.so libllamahip_c.so}; do
  echo "== $lib"
  export LLAMAHIP_LIB=$lib
  for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), [round(s['us_per_launch'],2) for s in d['roofline']['per_shape']])"; done
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pa && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o pa -- python /root/repo/bench.py --no-cpu-baseline > /dev/null 2>&1)
  python tools/prof_summary.py $(find /tmp/pa -name "*kernel_stats.csv") | grep "k_gemv<2, 2\|k_gemv<2, 0, 8\|k_gemv<0, 1, 10\|k_gemv<0, 1, 16\|k_dec_"
done
