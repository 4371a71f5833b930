cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in libmmt_hip.so libmmt_hip_w3_PLAINREADS.so libmmt_hip_w3_NOREADS.so libmmt_hip_w3_NOMFMA.so; do
MMT_HIP_LIB=mmt_amd/lib/$v timeout 200 python tools/wgrad3_lab.py 2>&1 | tail -1
done
