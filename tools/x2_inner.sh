mkdir -p gpurun_out/x2inner
for k in 1 2 4 8; do
  timeout -k 10 200 python3 tools/bench_x2.py $k --no-parity --blocks 40 > gpurun_out/x2inner/k$k.txt 2>&1 || exit 1
  python3 - gpurun_out/x2inner/k$k.txt $k <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][-1])
print(f"k={sys.argv[2]}: kernel median {d['kernel_ms']*1e3:.1f} us (min {d['kernel_ms_min']*1e3:.1f}), {d['value']:.4e} aircraft-steps/s")
PY
done
