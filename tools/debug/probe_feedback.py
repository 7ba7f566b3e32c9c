"""What does the vertex pipeline produce for the reference's own frame?  GL feedback mode returns window coordinates,
lit colours and texcoords of every (clipped) polygon as floats."""
from glprobe import *
import refshim_gl
env = refshim_gl.make_env(sys.argv[1] if len(sys.argv) > 1 else "Hallway")
env.reset(seed=int(sys.argv[2]) if len(sys.argv) > 2 else 0)
env.render_obs()
fb = env.obs_fb
fb.bind()
# same state as render_obs leaves behind: matrices are still loaded
mv = (c_float * 16)(); pr = (c_float * 16)()
gl.glGetFloatv(gl.GL_MODELVIEW_MATRIX, mv); gl.glGetFloatv(gl.GL_PROJECTION_MATRIX, pr)
print("MV", np.array(mv).reshape(4, 4).T)
print("P", np.array(pr).reshape(4, 4).T)
buf = (c_float * 100000)()
gl.glFeedbackBuffer(100000, gl.GL_4D_COLOR_TEXTURE, buf)
gl.glRenderMode(gl.GL_FEEDBACK)
gl.glCallList(1)
n = gl.glRenderMode(gl.GL_RENDER)
print("feedback floats", n)
a = np.array(buf[:n], np.float32)
i = 0
k = 0
while i < n and k < 12:
    tok = int(a[i]); i += 1
    if tok == gl.GL_POLYGON_TOKEN:
        nv = int(a[i]); i += 1
        print("poly", nv)
        for v in range(nv):
            rec = a[i:i + 12]; i += 12
            print("   xyzw", [float.hex(float(x)) for x in rec[:4]], "rgba", rec[4:8], "st", rec[8:10])
        k += 1
    else:
        print("token", hex(tok)); break
