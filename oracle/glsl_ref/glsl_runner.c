/*
 * glsl_runner — runs the REFERENCE's own GLSL compute shaders, unmodified, on the CPU
 * (Mesa llvmpipe) without any window system.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the "reference itself, run here" leg of the oracle (task rule 3): it is used in the
 * build container to (a) pin oracle/pt_oracle.c against the reference's real output and
 * (b) generate the golden fixtures committed under tests/golden/.  It reads the shader text
 * from /root/reference at run time (path given on the command line) — no reference source is
 * copied into this repository.  It never runs on the GPU box (no Mesa dependency travels).
 *
 * How: dlopen() Mesa's swrast_dri.so and drive it through the raw DRI "swrast loader"
 * interface (/usr/include/GL/internal/dri_interface.h) to obtain a GL 4.5 core context on
 * llvmpipe; then replay exactly what the reference host does around its dispatch:
 *   PathTracer.Render()            /root/reference/OpenTK-PathTracer/src/Render/PathTracer.cs:114-129
 *   UBO 0/1 allocation + binding   /root/reference/OpenTK-PathTracer/src/MainWindow.cs:195-201
 *   cubemap sampler state          /root/reference/OpenTK-PathTracer/src/MainWindow.cs:168,178
 *   AtmosphericScatterer.Render()  /root/reference/OpenTK-PathTracer/src/Render/AtmosphericScatterer.cs:63-113
 *
 * Job file (little endian), produced by tests/golden/make_golden.py:
 *   int32  magic 0x4a4c5347 ("GSLJ"), int32 mode (0 = path tracer, 1 = atmosphere)
 *   mode 0: int32 W,H,rayDepth,SPP,frameStart,numFrames,dumpEach,envSize,envFormat(0=RGBA32F,1=SRGB8_A8)
 *           float numSpheres,numCuboids,focalLength,apertureDiameter
 *           144 B BasicDataUBO, 26624 B GameObjectsUBO, 6 faces env (envSize^2 * (16|4) B each)
 *           output: (dumpEach ? numFrames : 1) * W*H*4 float32
 *   mode 1: int32 size,iSteps,jSteps ; float lightPos[3], lightIntensity ; 464 B AtmosphericDataUBO
 *           output: 6 * size*size*4 float32
 *   mode 2: int32 W,H ; W*H*4 float32 image.  Runs a compute shader (local size 8x8) that transforms the RGBA32F image
 *           bound at image unit 0 in place — used to drive the reference's post-process FUNCTIONS
 *           (PostProcessing/fragment.glsl:28-43) through a test main; output: W*H*4 float32
 *
 * Build: gcc -O2 -D_GNU_SOURCE glsl_runner.c -o ../_ref/glsl_runner -ldl -lm   (see oracle/Makefile)
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <GL/gl.h>
#include <GL/glext.h>
#include <GL/internal/dri_interface.h>

#define DIE(...) do { fprintf(stderr, "glsl_runner: " __VA_ARGS__); fputc('\n', stderr); exit(2); } while (0)

/* ---- swrast loader callbacks: we never present, so these are no-ops ---- */
static void cb_getDrawableInfo(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p)
{ (void)d; (void)p; *x = 0; *y = 0; *w = 16; *h = 16; }
static void cb_putImage(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void cb_getImage(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void cb_putImage2(__DRIdrawable *d, int op, int x, int y, int w, int h, int s, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)s; (void)data; (void)p; }
static void cb_getImage2(__DRIdrawable *d, int x, int y, int w, int h, int s, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)w; (void)h; (void)s; (void)data; (void)p; }

static const __DRIswrastLoaderExtension g_loader = {
    .base = { __DRI_SWRAST_LOADER, 3 },
    .getDrawableInfo = cb_getDrawableInfo,
    .putImage = cb_putImage,
    .getImage = cb_getImage,
    .putImage2 = cb_putImage2,
    .getImage2 = cb_getImage2,
};
static const __DRIextension *g_loader_exts[] = { &g_loader.base, NULL };

typedef void *(*getproc_t)(const char *);
static getproc_t g_getproc;
#define GLFN(type, name) type name = (type)g_getproc(#name); if (!name) DIE("missing GL entry point %s", #name)

static char *slurp(const char *path, size_t *len)
{
    FILE *f = fopen(path, "rb");
    if (!f) DIE("cannot open %s", path);
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = malloc((size_t)n + 1);
    if (fread(buf, 1, (size_t)n, f) != (size_t)n) DIE("short read on %s", path);
    buf[n] = 0;
    fclose(f);
    if (len) *len = (size_t)n;
    return buf;
}

static void make_context(void)
{
    const char *drv = getenv("GLSL_RUNNER_DRI");
    if (!drv) drv = "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so";
    void *h = dlopen(drv, RTLD_NOW | RTLD_GLOBAL);
    if (!h) DIE("dlopen %s: %s", drv, dlerror());
    const __DRIextension **(*get_exts)(void) =
        (const __DRIextension **(*)(void))dlsym(h, "__driDriverGetExtensions_swrast");
    if (!get_exts) DIE("no __driDriverGetExtensions_swrast");
    const __DRIextension **exts = get_exts();
    const __DRIcoreExtension *core = NULL;
    const __DRIswrastExtension *swrast = NULL;
    for (int i = 0; exts[i]; i++) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension *)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) swrast = (const __DRIswrastExtension *)exts[i];
    }
    if (!core || !swrast || swrast->base.version < 4) DIE("driver lacks DRI_Core / DRI_SWRast v4");
    const __DRIconfig **configs = NULL;
    __DRIscreen *screen = swrast->createNewScreen2(0, g_loader_exts, exts, &configs, NULL);
    if (!screen || !configs || !configs[0]) DIE("createNewScreen2 failed");
    uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 5 };
    unsigned err = 0;
    __DRIcontext *ctx = swrast->createContextAttribs(screen, __DRI_API_OPENGL_CORE, configs[0], NULL,
                                                     2, attribs, &err, NULL);
    if (!ctx) DIE("createContextAttribs failed (err %u)", err);
    __DRIdrawable *draw = swrast->createNewDrawable(screen, configs[0], NULL);
    if (!draw) DIE("createNewDrawable failed");
    if (!core->bindContext(ctx, draw, draw)) DIE("bindContext failed");
    g_getproc = (getproc_t)dlsym(RTLD_DEFAULT, "_glapi_get_proc_address");
    if (!g_getproc) DIE("no _glapi_get_proc_address");
}

static GLuint build_program(const char *src)
{
    GLFN(PFNGLCREATESHADERPROC, glCreateShader);
    GLFN(PFNGLSHADERSOURCEPROC, glShaderSource);
    GLFN(PFNGLCOMPILESHADERPROC, glCompileShader);
    GLFN(PFNGLGETSHADERIVPROC, glGetShaderiv);
    GLFN(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog);
    GLFN(PFNGLCREATEPROGRAMPROC, glCreateProgram);
    GLFN(PFNGLATTACHSHADERPROC, glAttachShader);
    GLFN(PFNGLLINKPROGRAMPROC, glLinkProgram);
    GLFN(PFNGLGETPROGRAMIVPROC, glGetProgramiv);
    GLFN(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog);
    /* The reference files start with a UTF-8 BOM on some shaders; GLSL wants '#version' first. */
    if ((unsigned char)src[0] == 0xEF && (unsigned char)src[1] == 0xBB && (unsigned char)src[2] == 0xBF) src += 3;
    const char *ver = strstr(src, "#version");
    if (ver) {
        /* comments before #version are legal GLSL; keep the text as is */
    }
    GLuint sh = glCreateShader(GL_COMPUTE_SHADER);
    glShaderSource(sh, 1, &src, NULL);
    glCompileShader(sh);
    GLint ok = 0;
    char log[8192];
    glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
    if (!ok) { glGetShaderInfoLog(sh, sizeof log, NULL, log); DIE("compile failed:\n%s", log); }
    GLuint prog = glCreateProgram();
    glAttachShader(prog, sh);
    glLinkProgram(prog);
    glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) { glGetProgramInfoLog(prog, sizeof log, NULL, log); DIE("link failed:\n%s", log); }
    return prog;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

struct reader { const unsigned char *p, *end; };
static const void *take(struct reader *r, size_t n)
{
    if ((size_t)(r->end - r->p) < n) DIE("job file truncated");
    const void *q = r->p;
    r->p += n;
    return q;
}
static int32_t take_i32(struct reader *r) { int32_t v; memcpy(&v, take(r, 4), 4); return v; }
static float take_f32(struct reader *r) { float v; memcpy(&v, take(r, 4), 4); return v; }

static void run_pathtracer(const char *shader_path, struct reader *r, FILE *out)
{
    int W = take_i32(r), H = take_i32(r), rayDepth = take_i32(r), spp = take_i32(r);
    int frameStart = take_i32(r), numFrames = take_i32(r), dumpEach = take_i32(r);
    int envSize = take_i32(r), envFormat = take_i32(r);
    float numSpheres = take_f32(r), numCuboids = take_f32(r), focal = take_f32(r), aperture = take_f32(r);
    const void *ubo0 = take(r, 144);
    const void *ubo1 = take(r, 26624);
    size_t texel = envFormat == 0 ? 16 : 4;
    size_t faceBytes = (size_t)envSize * envSize * texel;
    const void *faces[6];
    for (int f = 0; f < 6; f++) faces[f] = take(r, faceBytes);

    char *src = slurp(shader_path, NULL);
    GLuint prog = build_program(src);

    GLFN(PFNGLUSEPROGRAMPROC, glUseProgram);
    GLFN(PFNGLCREATEBUFFERSPROC, glCreateBuffers);
    GLFN(PFNGLNAMEDBUFFERSTORAGEPROC, glNamedBufferStorage);
    GLFN(PFNGLBINDBUFFERRANGEPROC, glBindBufferRange);
    GLFN(PFNGLCREATETEXTURESPROC, glCreateTextures);
    GLFN(PFNGLTEXTURESTORAGE2DPROC, glTextureStorage2D);
    GLFN(PFNGLTEXTURESUBIMAGE3DPROC, glTextureSubImage3D);
    GLFN(PFNGLTEXTURESUBIMAGE2DPROC, glTextureSubImage2D);
    GLFN(PFNGLTEXTUREPARAMETERIPROC, glTextureParameteri);
    GLFN(PFNGLBINDTEXTUREUNITPROC, glBindTextureUnit);
    GLFN(PFNGLBINDIMAGETEXTUREPROC, glBindImageTexture);
    GLFN(PFNGLGETUNIFORMLOCATIONPROC, glGetUniformLocation);
    GLFN(PFNGLPROGRAMUNIFORM1IPROC, glProgramUniform1i);
    GLFN(PFNGLPROGRAMUNIFORM1FPROC, glProgramUniform1f);
    GLFN(PFNGLPROGRAMUNIFORM2FPROC, glProgramUniform2f);
    GLFN(PFNGLDISPATCHCOMPUTEPROC, glDispatchCompute);
    GLFN(PFNGLMEMORYBARRIERPROC, glMemoryBarrier);
    GLFN(PFNGLGETTEXTUREIMAGEPROC, glGetTextureImage);
    void (*p_glEnable)(GLenum) = (void (*)(GLenum))g_getproc("glEnable");
    GLenum (*p_glGetError)(void) = (GLenum (*)(void))g_getproc("glGetError");
    void (*p_glFinish)(void) = (void (*)(void))g_getproc("glFinish");

    p_glEnable(GL_TEXTURE_CUBE_MAP_SEAMLESS); /* MainWindow.cs:168 */

    GLuint bufs[2];
    glCreateBuffers(2, bufs);
    glNamedBufferStorage(bufs[0], 144, ubo0, GL_DYNAMIC_STORAGE_BIT);
    glBindBufferRange(GL_UNIFORM_BUFFER, 0, bufs[0], 0, 144);
    glNamedBufferStorage(bufs[1], 26624, ubo1, GL_DYNAMIC_STORAGE_BIT);
    glBindBufferRange(GL_UNIFORM_BUFFER, 1, bufs[1], 0, 26624);

    GLuint env;
    glCreateTextures(GL_TEXTURE_CUBE_MAP, 1, &env);
    glTextureParameteri(env, GL_TEXTURE_MIN_FILTER, GL_NEAREST); /* MainWindow.cs:178 */
    glTextureParameteri(env, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
    glTextureStorage2D(env, 1, envFormat == 0 ? GL_RGBA32F : GL_SRGB8_ALPHA8, envSize, envSize);
    for (int f = 0; f < 6; f++)
        glTextureSubImage3D(env, 0, 0, 0, f, envSize, envSize, 1, GL_RGBA,
                            envFormat == 0 ? GL_FLOAT : GL_UNSIGNED_BYTE, faces[f]);

    GLuint img;
    glCreateTextures(GL_TEXTURE_2D, 1, &img);
    glTextureStorage2D(img, 1, GL_RGBA32F, W, H);
    float *zero = calloc((size_t)W * H * 4, sizeof(float));
    glTextureSubImage2D(img, 0, 0, 0, W, H, GL_RGBA, GL_FLOAT, zero);

    glProgramUniform2f(prog, glGetUniformLocation(prog, "uboGameObjectsSize"), numSpheres, numCuboids);
    glProgramUniform1i(prog, glGetUniformLocation(prog, "rayDepth"), rayDepth);
    glProgramUniform1i(prog, glGetUniformLocation(prog, "SPP"), spp);
    glProgramUniform1f(prog, glGetUniformLocation(prog, "focalLength"), focal);
    glProgramUniform1f(prog, glGetUniformLocation(prog, "apertureDiameter"), aperture);

    glUseProgram(prog);
    glBindTextureUnit(1, env);
    glBindImageTexture(0, img, 0, GL_FALSE, 0, GL_READ_WRITE, GL_RGBA32F);

    float *pix = zero;
    double t_total = 0.0;
    for (int f = 0; f < numFrames; f++) {
        glProgramUniform1i(prog, 0, frameStart + f); /* PathTracer.cs:117, location 0 */
        double t0 = now_s();
        glDispatchCompute((W + 7) / 8, (H + 7) / 8, 1); /* PathTracer.cs:121 */
        glMemoryBarrier(GL_ALL_BARRIER_BITS);
        p_glFinish();
        t_total += now_s() - t0;
        if (dumpEach || f == numFrames - 1) {
            glGetTextureImage(img, 0, GL_RGBA, GL_FLOAT, (GLsizei)((size_t)W * H * 16), pix);
            fwrite(pix, 16, (size_t)W * H, out);
        }
    }
    GLenum e = p_glGetError();
    if (e) DIE("glGetError = 0x%x", e);
    fprintf(stderr, "glsl_runner: pathtracer %dx%d depth %d spp %d frames %d: %.3f ms/frame, %.3f Msamples/s\n",
            W, H, rayDepth, spp, numFrames, 1e3 * t_total / numFrames,
            1e-6 * (double)W * H * spp * numFrames / t_total);
}

static void run_atmosphere(const char *shader_path, struct reader *r, FILE *out)
{
    int size = take_i32(r), iSteps = take_i32(r), jSteps = take_i32(r);
    float lp[3];
    for (int i = 0; i < 3; i++) lp[i] = take_f32(r);
    float intensity = take_f32(r);
    const void *ubo = take(r, 464);

    char *src = slurp(shader_path, NULL);
    GLuint prog = build_program(src);

    GLFN(PFNGLUSEPROGRAMPROC, glUseProgram);
    GLFN(PFNGLCREATEBUFFERSPROC, glCreateBuffers);
    GLFN(PFNGLNAMEDBUFFERSTORAGEPROC, glNamedBufferStorage);
    GLFN(PFNGLBINDBUFFERRANGEPROC, glBindBufferRange);
    GLFN(PFNGLCREATETEXTURESPROC, glCreateTextures);
    GLFN(PFNGLTEXTURESTORAGE2DPROC, glTextureStorage2D);
    GLFN(PFNGLBINDIMAGETEXTUREPROC, glBindImageTexture);
    GLFN(PFNGLGETUNIFORMLOCATIONPROC, glGetUniformLocation);
    GLFN(PFNGLPROGRAMUNIFORM1IPROC, glProgramUniform1i);
    GLFN(PFNGLPROGRAMUNIFORM1FPROC, glProgramUniform1f);
    GLFN(PFNGLPROGRAMUNIFORM3FPROC, glProgramUniform3f);
    GLFN(PFNGLDISPATCHCOMPUTEPROC, glDispatchCompute);
    GLFN(PFNGLMEMORYBARRIERPROC, glMemoryBarrier);
    GLFN(PFNGLGETTEXTURESUBIMAGEPROC, glGetTextureSubImage);
    GLenum (*p_glGetError)(void) = (GLenum (*)(void))g_getproc("glGetError");
    void (*p_glFinish)(void) = (void (*)(void))g_getproc("glFinish");

    GLuint buf;
    glCreateBuffers(1, &buf);
    glNamedBufferStorage(buf, 464, ubo, GL_DYNAMIC_STORAGE_BIT);
    glBindBufferRange(GL_UNIFORM_BUFFER, 2, buf, 0, 464); /* AtmosphericScatterer.cs:73 */

    GLuint cube;
    glCreateTextures(GL_TEXTURE_CUBE_MAP, 1, &cube);
    glTextureStorage2D(cube, 1, GL_RGBA32F, size, size);

    glProgramUniform3f(prog, glGetUniformLocation(prog, "lightPos"), lp[0], lp[1], lp[2]);
    glProgramUniform1f(prog, glGetUniformLocation(prog, "lightIntensity"), intensity);
    glProgramUniform1i(prog, glGetUniformLocation(prog, "iSteps"), iSteps);
    glProgramUniform1i(prog, glGetUniformLocation(prog, "jSteps"), jSteps);

    glUseProgram(prog);
    glBindImageTexture(0, cube, 0, GL_TRUE, 0, GL_WRITE_ONLY, GL_RGBA32F); /* AtmosphericScatterer.cs:106 */
    double t0 = now_s();
    glDispatchCompute((size + 7) / 8, (size + 7) / 8, 6); /* AtmosphericScatterer.cs:109 */
    glMemoryBarrier(GL_ALL_BARRIER_BITS);
    p_glFinish();
    double dt = now_s() - t0;
    size_t n = (size_t)size * size * 6 * 4;
    float *pix = malloc(n * sizeof(float));
    glGetTextureSubImage(cube, 0, 0, 0, 0, size, size, 6, GL_RGBA, GL_FLOAT, (GLsizei)(n * 4), pix);
    GLenum e = p_glGetError();
    if (e) DIE("glGetError = 0x%x", e);
    fwrite(pix, 4, n, out);
    fprintf(stderr, "glsl_runner: atmosphere %d^2 x6, %d x %d steps: %.3f ms\n", size, iSteps, jSteps, 1e3 * dt);
}

static void run_image_transform(const char *shader_path, struct reader *r, FILE *out)
{
    int W = take_i32(r), H = take_i32(r);
    const void *pixels = take(r, (size_t)W * H * 16);
    char *src = slurp(shader_path, NULL);
    GLuint prog = build_program(src);
    GLFN(PFNGLUSEPROGRAMPROC, glUseProgram);
    GLFN(PFNGLCREATETEXTURESPROC, glCreateTextures);
    GLFN(PFNGLTEXTURESTORAGE2DPROC, glTextureStorage2D);
    GLFN(PFNGLTEXTURESUBIMAGE2DPROC, glTextureSubImage2D);
    GLFN(PFNGLBINDIMAGETEXTUREPROC, glBindImageTexture);
    GLFN(PFNGLDISPATCHCOMPUTEPROC, glDispatchCompute);
    GLFN(PFNGLMEMORYBARRIERPROC, glMemoryBarrier);
    GLFN(PFNGLGETTEXTUREIMAGEPROC, glGetTextureImage);
    GLenum (*p_glGetError)(void) = (GLenum (*)(void))g_getproc("glGetError");
    void (*p_glFinish)(void) = (void (*)(void))g_getproc("glFinish");
    GLuint img;
    glCreateTextures(GL_TEXTURE_2D, 1, &img);
    glTextureStorage2D(img, 1, GL_RGBA32F, W, H);
    glTextureSubImage2D(img, 0, 0, 0, W, H, GL_RGBA, GL_FLOAT, pixels);
    glUseProgram(prog);
    glBindImageTexture(0, img, 0, GL_FALSE, 0, GL_READ_WRITE, GL_RGBA32F);
    glDispatchCompute((W + 7) / 8, (H + 7) / 8, 1);
    glMemoryBarrier(GL_ALL_BARRIER_BITS);
    p_glFinish();
    float *pix = malloc((size_t)W * H * 16);
    glGetTextureImage(img, 0, GL_RGBA, GL_FLOAT, (GLsizei)((size_t)W * H * 16), pix);
    GLenum e = p_glGetError();
    if (e) DIE("glGetError = 0x%x", e);
    fwrite(pix, 16, (size_t)W * H, out);
}

int main(int argc, char **argv)
{
    if (argc != 4) DIE("usage: glsl_runner <shader.glsl> <job.bin> <out.bin>");
    size_t n;
    unsigned char *job = (unsigned char *)slurp(argv[2], &n);
    struct reader r = { job, job + n };
    if (take_i32(&r) != 0x4a4c5347) DIE("bad job magic");
    int mode = take_i32(&r);
    make_context();
    FILE *out = fopen(argv[3], "wb");
    if (!out) DIE("cannot write %s", argv[3]);
    if (mode == 0) run_pathtracer(argv[1], &r, out);
    else if (mode == 1) run_atmosphere(argv[1], &r, out);
    else if (mode == 2) run_image_transform(argv[1], &r, out);
    else DIE("unknown mode %d", mode);
    fclose(out);
    return 0;
}
