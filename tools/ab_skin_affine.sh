mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_skinning.py tests/test_gpu_prologue.py tests/test_gpu_ztrajectory.py tests/test_gpu_zzbench_loop.py -m gpu -x -q 2>&1 | tail -8
for v in 1 0; do
  LAB4D_SKIN_AFFINE=$v timeout 600 python bench.py --gpus 1 --steps 4 --warmup 2 > gpurun_out/ab_skin_affine_$v.json 2> gpurun_out/ab_skin_affine_$v.err || tail -5 gpurun_out/ab_skin_affine_$v.err
  python - <<PY
import json
d=json.load(open('gpurun_out/ab_skin_affine_$v.json'))
print('SKIN_AFFINE=$v', d['value'], d['ms_per_step'], d['loss_last_chunk'], d['params_finite'], d['peak_hbm_gib'])
ks=d['roofline']['kernels_ms_per_step']
print({k:v for k,v in ks.items() if 'Skin' in k or 'bone' in k or 'blend' in k or 'wgrad_dma<2' in k or 'wgrad_dma<1' in k})
PY
done
