/* Headless OpenGL context on Mesa's software rasteriser (llvmpipe), TEST INFRASTRUCTURE ONLY.
 *
 * Gives tools/refshim_gl.py a real GL 4.5 compatibility context without an X server, EGL or OSMesa, so
 * that the reference's own render path (/root/reference/miniworld/opengl.py:202-435,
 * miniworld.py:1019-1221) runs unmodified and produces the pixels the oracle is pinned on.
 *
 * Recipe: dlopen libglapi (the dispatch table) and the swrast DRI driver, take the driver's
 * extension list, create a screen through DRI_SWRast v4 with a stub loader (the window-system
 * callbacks never matter: the reference draws into its own FBOs), create a compatibility context
 * and a dummy drawable, bind.  GL entry points come from _glapi_get_proc_address.
 *
 * Build: see oracle/Makefile (target _ref/libglctx.so).  Never linked into the product.
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <GL/gl.h>
#include <GL/internal/dri_interface.h>

static void *h_glapi, *h_drv;
static const __DRIcoreExtension *core;
static const __DRIswrastExtension *swrast;
static __DRIscreen *screen;
static __DRIcontext *context;
static __DRIdrawable *drawable;
static void *(*get_proc)(const char *);
static char errbuf[256];

static void ld_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p) {
    (void)d; (void)p; *x = 0; *y = 0; *w = 16; *h = 16;
}
static void ld_put(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p) {
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p;
}
static void ld_get(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p) {
    (void)d; (void)x; (void)y; (void)p; memset(data, 0, (size_t)w * h * 4);
}
static void ld_put2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *p) {
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p;
}
static void ld_get2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *p) {
    (void)d; (void)x; (void)y; (void)w; (void)p; memset(data, 0, (size_t)stride * h);
}

static const __DRIswrastLoaderExtension loader_ext = {
    .base = {__DRI_SWRAST_LOADER, 3},
    .getDrawableInfo = ld_info,
    .putImage = ld_put,
    .getImage = ld_get,
    .putImage2 = ld_put2,
    .getImage2 = ld_get2,
};
static const __DRIextension *loader_exts[] = {&loader_ext.base, NULL};

const char *glctx_error(void) { return errbuf; }

static int fail(const char *what) {
    snprintf(errbuf, sizeof errbuf, "%s%s%s", what, dlerror() ? ": " : "", "");
    return -1;
}

/* driver_dir: directory holding swrast_dri.so (NULL: the distribution's default). */
int glctx_create(const char *driver_dir) {
    if (context) return 0;
    char path[512];
    h_glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!h_glapi) return fail("dlopen libglapi.so.0");
    snprintf(path, sizeof path, "%s/swrast_dri.so", driver_dir ? driver_dir : "/usr/lib/x86_64-linux-gnu/dri");
    h_drv = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h_drv) return fail("dlopen swrast_dri.so");
    const __DRIextension **(*get_ext)(void) =
        (const __DRIextension **(*)(void))dlsym(h_drv, __DRI_DRIVER_GET_EXTENSIONS "_swrast");
    if (!get_ext) return fail("no __driDriverGetExtensions_swrast");
    const __DRIextension **exts = get_ext();
    for (int i = 0; exts[i]; i++) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension *)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) swrast = (const __DRIswrastExtension *)exts[i];
    }
    if (!core || !swrast || swrast->base.version < 4) return fail("driver lacks DRI_Core / DRI_SWRast v4");
    const __DRIconfig **configs = NULL;
    screen = swrast->createNewScreen2(0, loader_exts, exts, &configs, NULL);
    if (!screen || !configs) return fail("createNewScreen2");
    /* any RGBA8 + depth config will do: the reference renders into FBOs of its own */
    const __DRIconfig *cfg = NULL;
    for (int i = 0; configs[i]; i++) {
        unsigned r = 0, dep = 0, db = 0, smp = 0;
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_RED_SIZE, &r);
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_DEPTH_SIZE, &dep);
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_DOUBLE_BUFFER, &db);
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_SAMPLES, &smp);
        if (r == 8 && dep >= 16 && !db && !smp) { cfg = configs[i]; break; }
    }
    if (!cfg) cfg = configs[0];
    uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 2, __DRI_CTX_ATTRIB_MINOR_VERSION, 1};
    unsigned err = 0;
    context = swrast->createContextAttribs(screen, __DRI_API_OPENGL, cfg, NULL, 2, attribs, &err, NULL);
    if (!context) { snprintf(errbuf, sizeof errbuf, "createContextAttribs failed (%u)", err); return -1; }
    drawable = swrast->createNewDrawable(screen, cfg, NULL);
    if (!drawable) return fail("createNewDrawable");
    if (!core->bindContext(context, drawable, drawable)) return fail("bindContext");
    get_proc = (void *(*)(const char *))dlsym(h_glapi, "_glapi_get_proc_address");
    if (!get_proc) return fail("no _glapi_get_proc_address");
    return 0;
}

void *glctx_getproc(const char *name) {
    void *p = get_proc ? get_proc(name) : NULL;
    if (!p && h_glapi) p = dlsym(h_glapi, name);
    return p;
}

void glctx_destroy(void) {
    if (!context) return;
    core->unbindContext(context);
    core->destroyDrawable(drawable);
    core->destroyContext(context);
    core->destroyScreen(screen);
    context = NULL;
}

#ifdef GLCTX_MAIN
int main(void) {
    if (glctx_create(NULL)) { fprintf(stderr, "%s\n", glctx_error()); return 1; }
    const GLubyte *(*getString)(GLenum) = (const GLubyte *(*)(GLenum))glctx_getproc("glGetString");
    void (*getIntegerv)(GLenum, GLint *) = (void (*)(GLenum, GLint *))glctx_getproc("glGetIntegerv");
    GLint ms = 0;
    getIntegerv(0x8D57, &ms);
    printf("%s | %s | GL_MAX_SAMPLES %d\n", getString(GL_RENDERER), getString(GL_VERSION), ms);
    glctx_destroy();
    return 0;
}
#endif
