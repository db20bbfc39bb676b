#!/bin/bash
# tools/host_path.py --all-stats on the entries that showed alternating slow calls in one collection: the default
# harness (every transfer of its own through pinned memory) three times, then with --pageable-uploads (the harness's own
# uploads AND downloads through pageable temporaries: stale registrations of the runtime) twice, with the container's CPU
# quota and the cgroup's throttle counters around it.  Per-call lines are kept for every call slower than 1.3x the
# entry's fastest.
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)   hardware threads: $(nproc)"
grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
for mode in "" "" "" "--pageable-uploads" "--pageable-uploads"; do
  echo "== host_path.py --all-stats --only=double --only=fixed_batch_msm_host $mode"
  python tools/host_path.py --all-stats --only=double --only=fixed_batch_msm_host $mode 2>&1 | grep -v amdgpu.ids | cut -c1-330 | python -c "
import sys, re
block = []
for line in sys.stdin:
    if line.startswith('    call'):
        block.append(line)
        continue
    if line.startswith('ozk_'):
        ts = [float(re.search(r'call \d+:\s+([\d.]+) ms', l).group(1)) for l in block]
        lo = min(ts[1:]) if len(ts) > 1 else 0
        print(line, end='')
        for i, (t, l) in enumerate(zip(ts, block)):
            if i > 0 and t > 1.3 * lo:
                print(l, end='')
        block = []
"
done
grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
