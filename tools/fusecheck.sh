mkdir -p gpurun_out/r5f
AFC_ENV_LIB=$PWD/tools/bin/libafc_fuse.so timeout 300 python tools/fuse_dbg.py 16 1 2 2>&1 | tail -6
AFC_ENV_LIB=$PWD/tools/bin/libafc_fuse.so timeout 300 python tools/fuse_dbg.py 8 1024 2 2>&1 | tail -18 | cut -c1-100
for E in 1024 512 256; do timeout 300 python tools/ab_inproc.py --share-slab --envs $E tools/bin/libafc_fuse.so:AFC_FUSE_TAIL=0 tools/bin/libafc_fuse.so:AFC_FUSE_TAIL=1 2>&1 | tail -3 | head -2; done
