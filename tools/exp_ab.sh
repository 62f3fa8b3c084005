# same-box A/B of two builds of the library: the in-tree one against tools/perturb/libuvc_hip_$1.so (alternating, 3 rounds);
# further arguments go to bench.py (e.g. --model_type deit_small_patch16_224 --batch 256)
V=$1; shift
cd $GRAFT_REPO_ROOT
cp uvc_amd/libuvc_hip.so /tmp/good.so
for r in 1 2 3; do
  for v in good $V; do
    if [ $v = good ]; then cp /tmp/good.so uvc_amd/libuvc_hip.so; else cp tools/perturb/libuvc_hip_$v.so uvc_amd/libuvc_hip.so; fi
    python bench.py --no_cpu_baseline --steps 60 "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v',d['value'],d['ms_per_step'],d['kernel_ms_per_step_standalone_sum'])"
  done
done
cp /tmp/good.so uvc_amd/libuvc_hip.so
