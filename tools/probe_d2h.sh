#!/bin/bash
# tools/bin/probe_d2h under the runtime's copy switches; blit launches counted from a kernel trace, SDMA copies from the runtime's log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {
  echo "== $*"
  env "$@" tools/bin/probe_d2h 16 0
  env "$@" tools/bin/probe_d2h 16 1
  rm -rf /tmp/kt; env "$@" rocprofv3 --kernel-trace -f csv -d /tmp/kt -- tools/bin/probe_d2h 16 0 > /dev/null 2>&1
  echo "   blit launches (__amd_rocclr_copyBuffer) in a traced run: $(cat /tmp/kt/*/*kernel_trace.csv 2>/dev/null | grep -c copyBuffer)"
  echo "   'HSA Copy' lines in the runtime log: $(env "$@" AMD_LOG_LEVEL=4 tools/bin/probe_d2h 16 0 2>&1 | grep -c 'HSA Copy')"
}
run X=1
run GPU_FORCE_BLIT_COPY_SIZE=0
run HSA_ENABLE_SDMA=1
run HSA_ENABLE_SDMA=0
env AMD_LOG_LEVEL=4 tools/bin/probe_d2h 16 0 2>&1 | grep -i -E "copy|blit|sdma" | head -12
