#!/bin/bash
# HIP API calls issued between hipStreamBeginCapture and hipStreamEndCapture (anything but launches / event / wait calls is suspect)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export COMMARL_SKIP_CHILD_CASES=1 COMMARL_CAPTURE_MODE=relaxed
rocprofv3 --hip-runtime-trace --output-format csv -d /tmp/hiptrace -o t -- python3 -m pytest $ROOT/tests/test_hip_ppo_parity.py -m gpu -q -x -k "test_two_ppo or test_multi_tensor or test_sampler or sharded_multistream" > $ROOT/gpurun_out/trace_capture.log 2>&1
python3 - <<'PY' > $ROOT/gpurun_out/trace_capture_calls.txt
import csv, glob, collections
f = [x for x in glob.glob('/tmp/hiptrace/**/*hip_api_trace.csv', recursive=True)][0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
inside = False
cnt = collections.Counter()
seq = []
for r in rows:
    fn = r['Function']
    if 'BeginCapture' in fn:
        inside = True; cnt = collections.Counter(); continue
    if 'EndCapture' in fn:
        inside = False
        print('capture:', dict(cnt))
        continue
    if inside:
        cnt[fn] += 1
PY
cat $ROOT/gpurun_out/trace_capture_calls.txt | tail -20
