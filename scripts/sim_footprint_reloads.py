"""Developer tool (CPU only): footprint-reload statistics of the warp kernel for one cfg2 reference view, from the pinned
oracle's sample positions -- reload steps per (plane, edge) step for the kernel's scheme (8 pixels x RDB planes per wave) and for
alternatives (pixel blocks, wider windows).  Numbers quoted in DESIGN.md 4.1."""
import importlib, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
syn = importlib.import_module('3dvnet_amd.synthetic')
from oracle import pinned
inp = syn.make_costvolume_inputs('cfg2', n_ref=2, seed=1234 + 2)
K, R, t = inp['K'], inp['rotmats'], inp['tvecs']
edges = inp['edges']
d0, dd, D = inp['depth']; H, W = inp['img_size']; h, w = inp['plane_size']
Hf, Wf = inp['feat'].shape[2:]
_, P = pinned.camera_blocks(K, R, t)
ref = int(edges[0, 0])
X = pinned.world_points(K, R, t, ref, d0, dd, D, (H, W), (h, w))
srcs = edges[1][edges[0] == ref].tolist()
tot = {}
def count(name, v): tot[name] = tot.get(name, 0) + v
for src in srcs:
    ix, iy = pinned.sample_positions(X, P[src], (H, W), (Hf, Wf))
    ix = ix.clamp(-1, Wf).reshape(D, h * w).numpy(); iy = iy.clamp(-1, Hf).reshape(D, h * w).numpy()
    x0 = np.floor(ix).astype(int); y0 = np.floor(iy).astype(int)
    b = y0 * 1000 + x0                                  # footprint id per (plane, pixel)
    # current scheme: wave = 8 consecutive pixels, chunk of 8 planes; step reloads if ANY pixel's footprint differs from prev plane (first plane of chunk always)
    for RDB in (4, 8):
        ch = np.ones((D, h * w), bool); ch[1:] = b[1:] != b[:-1]; ch[::RDB] = True
        any8 = ch.reshape(D, -1, 8).any(axis=2)
        count('steps', any8.size) if RDB == 8 else None
        count('reload_steps_RDB%d' % RDB, int(any8.sum()))
        count('pixel_reloads_RDB%d' % RDB, int(ch.sum()))
    # pixel block 4x2
    chg = np.ones((D, h, w), bool); bb = b.reshape(D, h, w); chg[1:] = bb[1:] != bb[:-1]; chg[::8] = True
    blk = chg.reshape(D, h // 2, 2, w // 4, 4).any(axis=(2, 4))
    count('reload_steps_4x2', int(blk.sum()))
    blk24 = chg.reshape(D, h // 4, 4, w // 2, 2).any(axis=(2, 4))
    count('reload_steps_2x4', int(blk24.sum()))
    col = chg.reshape(D, h // 8, 8, w).any(axis=2)
    count('reload_steps_1x8', int(col.sum()))
    # 3x2 window anchored at even x: footprint id = (y0, x0 // 2 * 2) with special: x0 odd -> cells x0,x0+1 = anchor+1, anchor+2 ok
    a = y0 * 1000 + (x0 // 2)
    ch2 = np.ones((D, h * w), bool); ch2[1:] = a[1:] != a[:-1]; ch2[::8] = True
    count('reload_steps_win3x2', int(ch2.reshape(D, -1, 8).any(axis=2).sum()))
    a3 = (y0 // 2) * 1000 + (x0 // 2)
    ch3 = np.ones((D, h * w), bool); ch3[1:] = a3[1:] != a3[:-1]; ch3[::8] = True
    count('reload_steps_win3x3', int(ch3.reshape(D, -1, 8).any(axis=2).sum()))
    # motion direction stats
    dx = np.abs(np.diff(x0, axis=0)); dy = np.abs(np.diff(y0, axis=0))
    count('moves_x', int((dx > 0).sum())); count('moves_y', int((dy > 0).sum())); count('moves_big', int(((dx > 1) | (dy > 1)).sum()))
steps = tot['steps']
for k, v in tot.items(): print('%-24s %10d  %.3f per step' % (k, v, v / steps))
print('2x4 loads/step: %.2f' % (4 * tot['reload_steps_2x4'] / steps))
print('loads/step now (RDB8): %.2f ; RDB4: %.2f ; 4x2: %.2f ; 1x8: %.2f ; win3x2 (6 loads): %.2f ; win3x3 (9 loads): %.2f' % (
    4 * tot['reload_steps_RDB8'] / steps, 4 * tot['reload_steps_RDB4'] / steps, 4 * tot['reload_steps_4x2'] / steps,
    4 * tot['reload_steps_1x8'] / steps, 6 * tot['reload_steps_win3x2'] / steps, 9 * tot['reload_steps_win3x3'] / steps))
