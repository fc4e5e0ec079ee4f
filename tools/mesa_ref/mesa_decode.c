/* mesa_decode.c -- TEST INFRASTRUCTURE, fixture generation only (never shipped, never on the
 * product path).  An INDEPENDENT block decoder: Mesa 23.2.1's software texture decompression
 * (src/mesa/main/texcompress_{astc,etc,bptc,s3tc,rgtc}) reached through an off-screen llvmpipe
 * OpenGL 4.5 context.  The image has no X server, EGL or OSMesa, so the context is created
 * straight on the DRI software-rasteriser interface of /usr/lib/x86_64-linux-gnu/dri/swrast_dri.so
 * (GL/internal/dri_interface.h: DRI_Core + DRI_SWRast with a do-nothing DRI_SWRastLoader), the
 * way libGLX's drisw loader does.  A compressed payload goes in through
 * glCompressedTexImage2D and comes back decoded through glGetTexImage.
 *
 * mesa_encode goes the other way: an INDEPENDENT ENCODER.  Pixels go in through glTexImage2D with a
 * compressed internal format -- Mesa's software drivers then compress on the CPU with their own
 * encoders (S3TC: the former libtxc_dxtn in texcompress_s3tc_tmp.h, RGTC: texcompress_rgtc,
 * BPTC: texcompress_bptc_tmp.h) -- and the payload comes back through glGetCompressedTexImage.
 *
 * Build:  gcc -O1 -shared -fPIC -o libmesa_decode.so mesa_decode.c -ldl
 * Used by tests/golden/make_mesa_fixtures.py (commits random valid blocks + Mesa's pixels) and,
 * when the driver file is present, by tests/test_mesa_crosscheck.py on live encoder output. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <dlfcn.h>
#include <stdint.h>
#include <GL/gl.h>
#include <GL/glext.h>
#include <GL/internal/dri_interface.h>

static void cb_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p)
{ (void)d; (void)p; *x = 0; *y = 0; *w = 16; *h = 16; }
static void cb_put(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void cb_get(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)p; memset(data, 0, (size_t)w * h * 4); }
static void cb_put2(__DRIdrawable *d, int op, int x, int y, int w, int h, int s, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)s; (void)data; (void)p; }
static void cb_get2(__DRIdrawable *d, int x, int y, int w, int h, int s, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)w; (void)p; memset(data, 0, (size_t)s * h); }

static const __DRIswrastLoaderExtension loader = {
    { __DRI_SWRAST_LOADER, 3 }, cb_info, cb_put, cb_get, cb_put2, cb_get2, 0, 0, 0, 0 };
static const __DRIextension *loader_exts[] = { &loader.base, NULL };

static int g_ready = 0;
static void *(*g_gpa)(const char *);
static void (*p_GenTextures)(GLsizei, GLuint *);
static void (*p_BindTexture)(GLenum, GLuint);
static void (*p_DeleteTextures)(GLsizei, const GLuint *);
static void (*p_TexParameteri)(GLenum, GLenum, GLint);
static void (*p_PixelStorei)(GLenum, GLint);
static void (*p_CompressedTexImage2D)(GLenum, GLint, GLenum, GLsizei, GLsizei, GLint, GLsizei, const void *);
static void (*p_GetTexImage)(GLenum, GLint, GLenum, GLenum, void *);
static void (*p_TexImage2D)(GLenum, GLint, GLint, GLsizei, GLsizei, GLint, GLenum, GLenum, const void *);
static void (*p_GetCompressedTexImage)(GLenum, GLint, void *);
static void (*p_GetTexLevelParameteriv)(GLenum, GLint, GLenum, GLint *);
static GLenum (*p_GetError)(void);
static const GLubyte *(*p_GetString)(GLenum);
static void (*p_Finish)(void);

const char *mesa_version(void) { return g_ready ? (const char *)p_GetString(GL_VERSION) : ""; }

int mesa_init(const char *driver_path)
{
    if (g_ready) return 0;
    void *glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!glapi) return -1;
    void *drv = dlopen(driver_path ? driver_path : "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so",
                       RTLD_NOW | RTLD_GLOBAL);
    if (!drv) return -2;
    const __DRIextension **(*getext)(void) =
        (const __DRIextension **(*)(void))dlsym(drv, "__driDriverGetExtensions_swrast");
    if (!getext) return -3;
    const __DRIextension **exts = getext();
    const __DRIcoreExtension *core = NULL;
    const __DRIswrastExtension *sw = NULL;
    for (int i = 0; exts[i]; i++) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension *)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) sw = (const __DRIswrastExtension *)exts[i];
    }
    if (!core || !sw || sw->base.version < 4) return -4;
    const __DRIconfig **configs = NULL;
    __DRIscreen *scr = sw->createNewScreen2(0, loader_exts, exts, &configs, NULL);
    if (!scr || !configs || !configs[0]) return -5;
    unsigned err = 0;
    uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, 3, __DRI_CTX_ATTRIB_MINOR_VERSION, 0 };
    __DRIcontext *ctx = sw->createContextAttribs(scr, __DRI_API_OPENGL, configs[0], NULL, 2, attribs, &err, NULL);
    if (!ctx) return -6;
    __DRIdrawable *dr = sw->createNewDrawable(scr, configs[0], NULL);
    if (!dr) return -7;
    if (!core->bindContext(ctx, dr, dr)) return -8;
    g_gpa = (void *(*)(const char *))dlsym(glapi, "_glapi_get_proc_address");
    if (!g_gpa) return -9;
#define GP(n) do { *(void **)&p_##n = g_gpa("gl" #n); if (!p_##n) return -10; } while (0)
    GP(GenTextures); GP(BindTexture); GP(DeleteTextures); GP(TexParameteri); GP(PixelStorei);
    GP(CompressedTexImage2D); GP(GetTexImage); GP(GetError); GP(GetString); GP(Finish);
    GP(TexImage2D); GP(GetCompressedTexImage); GP(GetTexLevelParameteriv);
#undef GP
    g_ready = 1;
    return 0;
}

/* payload of a w x h texture in GL internal format `glfmt` -> pixels read back as
 * (rb_format, rb_type), tightly packed, top row first.  Returns 0 or the GL error. */
int mesa_decode(unsigned glfmt, int w, int h, const void *data, int size,
                unsigned rb_format, unsigned rb_type, void *out)
{
    if (!g_ready) return -1;
    while (p_GetError() != GL_NO_ERROR) {}
    GLuint tex = 0;
    p_GenTextures(1, &tex);
    p_BindTexture(GL_TEXTURE_2D, tex);
    p_TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    p_TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    p_TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAX_LEVEL, 0);
    p_PixelStorei(GL_PACK_ALIGNMENT, 1);
    p_PixelStorei(GL_UNPACK_ALIGNMENT, 1);
    p_CompressedTexImage2D(GL_TEXTURE_2D, 0, glfmt, w, h, 0, size, data);
    int e = (int)p_GetError();
    if (!e) {
        p_GetTexImage(GL_TEXTURE_2D, 0, rb_format, rb_type, out);
        p_Finish();
        e = (int)p_GetError();
    }
    p_BindTexture(GL_TEXTURE_2D, 0);
    p_DeleteTextures(1, &tex);
    return e;
}

/* pixels (src_format, src_type; tightly packed, top row first) of a w x h texture -> the payload
 * Mesa's own software encoder produces for GL internal format `glfmt`.  Returns the payload size
 * (<= cap) or a negative GL error / -1. */
int mesa_encode(unsigned glfmt, int w, int h, unsigned src_format, unsigned src_type, const void *pixels,
                void *out, int cap)
{
    if (!g_ready) return -1;
    while (p_GetError() != GL_NO_ERROR) {}
    GLuint tex = 0;
    int size = -1;
    p_GenTextures(1, &tex);
    p_BindTexture(GL_TEXTURE_2D, tex);
    p_TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    p_TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    p_TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAX_LEVEL, 0);
    p_PixelStorei(GL_PACK_ALIGNMENT, 1);
    p_PixelStorei(GL_UNPACK_ALIGNMENT, 1);
    p_TexImage2D(GL_TEXTURE_2D, 0, (GLint)glfmt, w, h, 0, src_format, src_type, pixels);
    int e = (int)p_GetError();
    if (!e) {
        GLint compressed = 0, csize = 0, ifmt = 0;
        p_GetTexLevelParameteriv(GL_TEXTURE_2D, 0, GL_TEXTURE_COMPRESSED, &compressed);
        p_GetTexLevelParameteriv(GL_TEXTURE_2D, 0, GL_TEXTURE_COMPRESSED_IMAGE_SIZE, &csize);
        p_GetTexLevelParameteriv(GL_TEXTURE_2D, 0, GL_TEXTURE_INTERNAL_FORMAT, &ifmt);
        if (!compressed || (unsigned)ifmt != glfmt || csize <= 0 || csize > cap)
            e = 1;
        else {
            p_GetCompressedTexImage(GL_TEXTURE_2D, 0, out);
            p_Finish();
            e = (int)p_GetError();
            size = csize;
        }
    }
    p_BindTexture(GL_TEXTURE_2D, 0);
    p_DeleteTextures(1, &tex);
    return e ? -(e > 0 ? e : 1) : size;
}
