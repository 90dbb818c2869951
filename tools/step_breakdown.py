#!/usr/bin/env python
"""Per-geometry breakdown of the convolution launches of one DAFNet iteration (HIP events), sorted by time."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

recs = []
def install():
    from multimodal_segmentation_amd import _native
    orig = _native.call
    def call(name, *args):
        if name in ('mmseg_conv2d_fwd', 'mmseg_conv2d_wgrad', 'mmseg_conv2d_dgrad_parity') and recs is not None and install.on:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); rc = orig(name, *args); e.record()
            if name == 'mmseg_conv2d_fwd':
                g = args[7:22]
            elif name == 'mmseg_conv2d_wgrad':
                g = args[6:20]
            else:
                g = args[3:15]
            recs.append((name, tuple(g), s, e))
            return rc
        return orig(name, *args)
    _native.call = call
install.on = False

if __name__ == '__main__':
    sys.argv = [sys.argv[0], '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-conv-timer']
    install()
    import types
    # run bench.main with our hook switched on for all steps (timing includes warm-up: we divide by 3)
    install.on = True
    bench.main()
    torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0.0, 0])
    for name, g, s, e in recs:
        a = agg[(name, g)]
        a[0] += s.elapsed_time(e); a[1] += 1
    tot = sum(v[0] for v in agg.values())
    print('total conv ms per iteration: %.1f' % (tot / 3))
    for (name, g), (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
        print('%-26s %-70s n/it %5.1f  ms/it %6.2f  avg_us %7.1f' % (name.replace('mmseg_conv2d_', ''), str(g), n / 3, ms / 3, 1000 * ms / n))
