#!/bin/bash
# GPU box: where do kernel arguments live?  The step kernel's waves begin with scalar loads of CwParams from the kernarg segment; HIP_FORCE_DEV_KERNARG
# decides whether that segment is in device memory or in host memory behind PCIe.  State-only and full-frame steps with the variable unset / 1 / 0.
cd $GRAFT_REPO_ROOT
for mode in state pixels; do for v in unset 1 0; do
  if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  python bench.py --quick --obs-mode $mode --steps 600 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']
print('$mode HIP_FORCE_DEV_KERNARG=$v  ms_per_step %.5f  step kernel %.5f ms  sweep %s' % (d['ms_per_step'], k['step'] or 0, k['render']))"
done; done
