#!/bin/bash
# round 5, call V: cross-attention reverse-pass kernel — parity, reverse-pass suite, configs[3] line (guidance iteration eager / graphed) with and without the recompute kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5v; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q -k "reverse_pass" > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 1500 python -m pytest tests/test_backward_gpu.py tests/test_round3_gpu.py -x -q > $O/tests_bwd.log 2>&1; tail -4 $O/tests_bwd.log
timeout 900 python bench.py --plan sd21 --ddim-steps 10 --steps 1 --warmup 1 > $O/sd21_flash.json 2> $O/sd21_flash.err
TG_FLASH_BWD=0 timeout 900 python bench.py --plan sd21 --ddim-steps 10 --steps 1 --warmup 1 > $O/sd21_mat.json 2> $O/sd21_mat.err
python - <<'PY'
import json
for n in ("flash", "mat"):
    try:
        r = json.loads(open(f"gpurun_out/r5v/sd21_{n}.json").read().strip().splitlines()[-1])
        print(n, {k: r.get(k) for k in ("value", "latent_backward_guidance_iteration_ms", "latent_backward_guidance_iteration_graph_ms", "latent_backward_guidance_graph_note")})
    except Exception as e:
        print(n, "ERR", e)
PY
