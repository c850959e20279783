# developer tool (gpurun): goldie x 64 streams under different split-K policies of the multi-token GEMM
ulimit -c 0
cd $GRAFT_REPO_ROOT
for cfg in "1048576 16 1024" "6 16 1024" "4 16 1024" "3 16 1024" "2 16 1024" "3 32 2048" "2 32 2048" "1 32 2048"; do
  set -- $cfg
  echo "max_chunks=$1 ks_cap=$2 max_wg=$3: $(NL_QG_MAX_CHUNKS=$1 NL_KS_CAP=$2 NL_QG_MAX_WG=$3 python3 -c "
import sys; sys.path.insert(0,'tools'); import bench_modes as b; b.batch('goldie','q4_0',64,steps=24)" 2>&1 | tail -1)"
done
