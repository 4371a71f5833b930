cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c22
mkdir -p $O
cd $R
run() { # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 200 --warmup 15 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/$n.json
}
run base A=1
run kernarg1 HIP_FORCE_DEV_KERNARG=1
run kernarg0 HIP_FORCE_DEV_KERNARG=0
run pktcap1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run hwq1 GPU_MAX_HW_QUEUES=1
run nosdma HSA_ENABLE_SDMA=0
run base2 A=1
