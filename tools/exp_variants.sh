cd $GRAFT_REPO_ROOT
cp uvc_amd/libuvc_hip.so /tmp/good.so
for v in good e1 e2; do
  if [ $v = good ]; then cp /tmp/good.so uvc_amd/libuvc_hip.so; else cp tools/perturb/libuvc_hip_$v.so uvc_amd/libuvc_hip.so; fi
  echo "== $v"; python tools/gemm_sweep.py 2>&1 | grep -v amdgpu.ids | grep "N= 2304"
done
cp /tmp/good.so uvc_amd/libuvc_hip.so
