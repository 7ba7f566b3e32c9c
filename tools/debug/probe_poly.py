"""how does the real (non-feedback) path split an immediate-mode GL_QUADS quad, and in which vertex order is each half set up?"""
from probe_z import *
import itertools
def plane_for(v3):      # v3: three window verts in the order handed to setup (before the front-face swap)
    return planes(np.array(v3))
rng = np.random.default_rng(3)
for k in range(4):
    gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, ms)
    reset_state(); gl.glEnable(gl.GL_MULTISAMPLE); gl.glEnable(gl.GL_DEPTH_TEST)
    gl.glClearColor(0, 0, 0, 0); gl.glClearDepth(1.0); gl.glClear(gl.GL_COLOR_BUFFER_BIT | gl.GL_DEPTH_BUFFER_BIT)
    # a planar convex ccw quad in clip space: take a parallelogram in window space with a perspective w
    c = rng.uniform([8, 6], [W - 8, H - 6]); a = rng.uniform(3, 7, 2); b = np.array([-a[1], a[0]]) * rng.uniform(0.6, 1.2)
    p = np.array([c - a - b, c + a - b, c + a + b, c - a + b])
    wq = np.array([1.0, 1.7, 2.9, 2.2], np.float32) * rng.uniform(0.8, 1.5)
    # make it planar in clip space: z/w affine in window position
    zndc = 0.2 + 0.01 * p[:, 0] - 0.013 * p[:, 1]
    ndc = np.stack([p[:, 0] / W * 2 - 1, p[:, 1] / H * 2 - 1], axis=1)
    clip = np.zeros((4, 4), np.float32)
    clip[:, 0] = (ndc[:, 0] * wq).astype(np.float32); clip[:, 1] = (ndc[:, 1] * wq).astype(np.float32)
    clip[:, 2] = (zndc * wq).astype(np.float32); clip[:, 3] = wq
    gl.glBegin(gl.GL_POLYGON)
    for i in range(4):
        gl.glColor4f(1, 1, 1, 1); gl.glVertex4f(*[float(x) for x in clip[i]])
    gl.glEnd(); gl.glFlush()
    gl.glBindFramebuffer(gl.GL_READ_FRAMEBUFFER, ms); gl.glBindFramebuffer(gl.GL_DRAW_FRAMEBUFFER, ss)
    gl.glBlitFramebuffer(0, 0, W, H, 0, 0, W, H, gl.GL_DEPTH_BUFFER_BIT, gl.GL_NEAREST)
    gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, ss)
    z = np.zeros((H, W), np.float32)
    gl.glReadPixels(0, 0, W, H, gl.GL_DEPTH_COMPONENT, gl.GL_FLOAT, z.ctypes.data)
    win = np.zeros((4, 4), np.float32)
    for i in range(4):
        oow = f32(f32(1) / clip[i, 3])
        win[i, 0] = fma(f32(clip[i, 0] * oow), f32(W / 2), f32(W / 2))
        win[i, 1] = fma(f32(clip[i, 1] * oow), f32(H / 2), f32(H / 2))
        win[i, 2] = fma(f32(clip[i, 2] * oow), f32(0.5), f32(0.5)); win[i, 3] = oow
    cov = z < 1.0
    sx, sy = f32(0.375), f32(0.125)
    print("quad", k, "covered", int(cov.sum()))
    for tri in itertools.permutations(range(4), 3):
        # orientation must be ccw (same as quad order): check cyclic order
        a0, a1, a2 = tri
        cyc = [(a1 - a0) % 4, (a2 - a1) % 4, (a0 - a2) % 4]
        if sum(cyc) != 4: continue
        c0, dadx, dady = plane_for([win[a0], win[a1], win[a2]])
        n = 0
        for y, x in zip(*np.nonzero(cov)):
            v = fma(dady, f32(y) + sy, fma(dadx, f32(x) + sx, c0))
            n += int(v == z[y, x])
        if n > 5: print("   tri", tri, "exact pixels", n)
print("---- pair test on the last quad")
def side(pa, pb, x, y): return (pb[0] - pa[0]) * (y - pa[1]) - (pb[1] - pa[1]) * (x - pa[0])
best = []
for diag in ((1, 3), (0, 2)):
    if diag == (1, 3): tris = [(0, 1, 3), (1, 2, 3)]
    else: tris = [(0, 1, 2), (0, 2, 3)]
    rots = lambda t: [t, (t[1], t[2], t[0]), (t[2], t[0], t[1])]
    for ta in rots(tris[0]):
        for tb in rots(tris[1]):
            pa_, pb_ = plane_for([win[i] for i in ta]), plane_for([win[i] for i in tb])
            bad = 0
            for y, x in zip(*np.nonzero(cov)):
                xs, ys = f32(x) + sx, f32(y) + sy
                # which half: sign of the diagonal edge function relative to vertex tris[0][1 or so]
                d0, d1 = win[diag[0]], win[diag[1]]
                sA = side(d0, d1, win[tris[0][0] if tris[0][0] not in diag else tris[0][1]][0], win[tris[0][0] if tris[0][0] not in diag else tris[0][1]][1])
                sP = side(d0, d1, float(xs), float(ys))
                pl = pa_ if (sP > 0) == (sA > 0) else pb_
                v = fma(pl[2], ys, fma(pl[1], xs, pl[0]))
                bad += int(v != z[y, x])
            best.append((bad, diag, ta, tb))
best.sort()
for b in best[:6]: print(b)
