# developer tool (gpurun): goldie x 64 streams under different split-K policies / workgroup heights of the multi-token GEMM
ulimit -c 0
cd $GRAFT_REPO_ROOT
for cfg in ${CFGS:-"0 6 16 1024" "82 6 16 1024" "82 3 16 1024" "82 2 16 1024" "82 1 32 1024" "82 2 32 2048" "82 1 16 1024"}; do
  set -- $cfg
  echo "geom=$1 max_chunks=$2 ks_cap=$3 max_wg=$4: $(NL_QG_DEC_GEOM=$1 NL_QG_MAX_CHUNKS=$2 NL_KS_CAP=$3 NL_QG_MAX_WG=$4 python3 -c "
import sys; sys.path.insert(0,'tools'); import bench_modes as b; b.batch('goldie','q4_0',64,steps=24)" 2>&1 | tail -1)"
done
