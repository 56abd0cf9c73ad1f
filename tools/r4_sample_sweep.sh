run() { python bench.py --no-verify --no-cpu-baseline --no-peaks --no-secondary --steps $S --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); ro=r['roofline']; print(r['value'], r['ms_per_step'], ro['avg_launch_ms'], ro.get('sample_pass_avg_ms'), ro.get('finalize_avg_ms'))"; }
for shape in "" "--batch 256" "--config 1" "--dtype f32" "--metric l2" "--batch 64" "--batch 32"; do
  for div in 16 12 8 6; do
    S=60; [ "$shape" = "--config 1" ] && S=300
    echo -n "[$shape] div $div: "; run $shape --debug sample_div=$div
  done
done
