cd $GRAFT_REPO_ROOT
for rep in 1 2; do for mode in 1 0; do echo "dma=$mode"; MI_PLANES_DMA=$mode MI_PLANES_LATENCY=$([ $mode = 1 ] && echo 256 || echo 0) python bench.py --steps 60 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-130; done; done
