import sys, os
sys.path[:0] = ['/root/repo/tools', '/root/repo/tests', '/root/repo/flu-plugins-oss_amd']
import numpy as np, gstenv, oracle_lib as O
ROOT = '/root/repo'
z = np.load(ROOT + '/tests/golden/scenes/k06_animation.npz')
W, H = int(z['width']), int(z['height'])
ttml = ROOT + '/tests/golden/scenes/k06_animation.ttml'
times = gstenv.buffer_times(gstenv.HIP_PLUGIN_DIR, ttml, W, H)
print(times)
fps, n = 4, 15
for it in range(6):
    vraw, oraw = gstenv.render_over_video(ttml, W, H, n, fps)
    video = np.frombuffer(vraw, np.uint32).reshape(n, H, W)
    out = np.frombuffer(oraw, np.uint32).reshape(n, H, W)
    msg = []
    for k in range(n):
        t = k * 10**9 // fps
        st = [i for i, (p, d) in enumerate(times) if p <= t and (d is None or t < p + d)]
        match = [i for i in range(int(z['n_states'])) if np.array_equal(
            out[k], O.over(z['frames'][i].reshape(-1), video[k].reshape(-1)).reshape(H, W))]
        if not st or st[0] not in match:
            exp = O.over(z['frames'][st[0]].reshape(-1), video[k].reshape(-1)).reshape(H, W)
            ys, xs = np.nonzero(out[k] != exp)
            msg.append((k, st, match, (int(xs.min()), int(ys.min()), int(xs.max()), int(ys.max())), len(xs)))
            if k == 8:
                np.save(ROOT + '/gpurun_out/vp_out8.npy', out[k]); np.save(ROOT + '/gpurun_out/vp_exp8.npy', exp)
    print(it, msg or 'ok', flush=True)
