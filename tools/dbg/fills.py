"""Which torch ops launch fill kernels inside one training step of the headline workload?"""
import sys, collections, torch
sys.path.insert(0, '.')
sys.argv = ['bench.py'] + sys.argv[1:]
import bench  # noqa
from torch.profiler import profile, ProfilerActivity

orig = torch.cuda.synchronize
state = {'n': 0}
import argparse
# run bench's own main with a hook: profile the last of a few steps
def run():
    a = ['--steps', '2', '--warmup', '2', '--no-cpu-baseline', '--no-kernel-events'] + sys.argv[1:]
    sys.argv = ['bench.py'] + a
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
        bench.main()
    cnt = collections.Counter()
    for ev in prof.events():
        if ev.name in ('aten::zero_', 'aten::fill_', 'aten::zeros', 'aten::zeros_like', 'aten::ones', 'aten::full', 'aten::new_zeros'):
            st = [s for s in (ev.stack or []) if 'boxsegliver_amd' in s or 'bench.py' in s or 'autograd' in s]
            cnt[(ev.name, str(ev.input_shapes)[:60], (st[0] if st else (ev.stack[0] if ev.stack else '?'))[-110:])] += 1
    for k, v in cnt.most_common(40):
        print(v, k)
run()
