mkdir -p gpurun_out/r5f
for E in 256 64 1 768 1024; do timeout 300 python tools/ab_inproc.py --share-slab --envs $E tools/bin/libafc_fuse.so:AFC_FUSE_TAIL=0 tools/bin/libafc_fuse.so:AFC_FUSE_TAIL=1 tools/bin/libafc_fuse.so:AFC_FUSE_TAIL=2 2>&1 | tail -4 | head -3; done
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
for k in 1 2; do timeout 300 python bench.py --no-cpu-baseline --steps 64 2>/dev/null | tail -1 | cut -c1-330; done
