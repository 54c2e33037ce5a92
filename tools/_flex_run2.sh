mkdir -p gpurun_out/flex2
timeout 1800 python -m pytest tests -m gpu -q -k "flex or dopri_matches_oracle_small" 2>&1 | tail -60 | tee gpurun_out/flex2/gpu_flex.txt
