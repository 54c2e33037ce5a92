#!/bin/bash
# The profiles behind the bench line, all four workloads (headline + the three `secondary` ones), in one GPU-box call:
#   gpurun -- 'bash tools/gpu_profiles.sh r05'        ->  gpurun_out/<tag>_{quad_full,con,atlas,atlas_con}/summary/
# then, back in the repository:   for d in gpurun_out/<tag>_*/summary; do cp $d/* profiles/; done
# Each workload = tools/gpu_profile.sh: the bench line, `rocprofv3 --kernel-trace --stats`, then one `--pmc` pass per counter
# group (never combined with other trace domains), summarised on the box.  The pmc_*_latest.json files it writes are what
# bench.py quotes for `roofline.traffic` / the VALU-issue roof -- bench.py ignores them when model, batch, dtype or the
# extra-terms setting of the profiled run differ from its own, so a stale file can not leak into a bench line.
set -u
exec < /dev/null
TAG=${1:-r05}
bash tools/gpu_profile.sh ${TAG}_quad_full pmc_latest.json --no-secondary 2>&1 | tail -n 8 | cut -c1-200
JM_PROFILE_DOMINANT='%k_qcon_pgs_lane%' JM_PROFILE_KERNELS='%k_quad_con_split<%1_ 0>%,%k_quad_con_split<%2_ 0>%,%k_quad_con<%' \
  bash tools/gpu_profile.sh ${TAG}_con pmc_con_latest.json --no-secondary --contact-model constraint --steps 40 --warmup 5 2>&1 | tail -n 4 | cut -c1-200
bash tools/gpu_profile.sh ${TAG}_atlas pmc_atlas_latest.json --no-secondary --model atlas --batch 32768 --dt 2.5e-4 --steps 40 --warmup 5 2>&1 | tail -n 4 | cut -c1-200
JM_PROFILE_DOMINANT='%k_qtip_pgs%' JM_PROFILE_KERNELS='%k_quad_con_split<%1_ 0>%,%k_quad_con_split<%2_ 0>%,%k_qcon_pgs<%8,%,%k_qtip_exact%' \
  bash tools/gpu_profile.sh ${TAG}_atlas_con pmc_atlas_con_latest.json --no-secondary --model atlas --batch 32768 --contact-model constraint --dt 5e-4 --steps 20 --warmup 3 2>&1 | tail -n 4 | cut -c1-200
ls gpurun_out/${TAG}_*/summary
