#!/bin/bash
O=gpurun_out/r3j
mkdir -p $O
for lib in d-vae_amd/csrc/libdvae_hip.so; do
  echo "== $lib"
  DVAE_EXP_LIB=$lib timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -q -m gpu -k "hs501_matches_reference_golden" --timeout 120 --timeout-method thread -p no:cacheprovider 2>&1 | grep -E "passed|failed|AssertionError" | head -8
  python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/parity/grad_errors_*.json")):
    d=json.load(open(f))
    w=max(d.items(), key=lambda kv: kv[1]["hip_vs_exact"] if isinstance(kv[1],dict) and "hip_vs_exact" in kv[1] else -1)
    print(f.split("/")[-1], w[0], w[1])
PY
done
