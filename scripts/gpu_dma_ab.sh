cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_gemm.py -m gpu -x -q -k "latency" 2>&1 | tail -5
python scripts/small_gemm_probe.py 2>&1 | grep "256:"
python scripts/small_batch_step.py 10,7,4,10 200
python scripts/default_batch_streams.py | tail -4
for mode in 1 2 0; do MI_PLANES_DMA=$mode python bench.py --steps 20 --warmup 3 2>/dev/null | cut -c1-140; done
