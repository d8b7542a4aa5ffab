// pt_host.hpp — C++ host side above the C ABI, mirroring the reference's C# classes for this path (the reference
// host is compiled code and no C#/.NET toolchain exists in this image, so the mirror is C++17, header-only).
//
//   Material / Sphere / Cuboid / BaseSTD140Compatible   src/Material.cs, src/GameObjects/*.cs, src/BaseSTD140Compatible.cs
//   Camera (+ the OpenTK Matrix4 helpers it calls)      src/Camera.cs
//   UniformBuffer (BufferObject.SubData on UBO 0/1)     src/Render/Objects/BufferObject.cs:37-48
//   PathTracer                                          src/Render/PathTracer.cs:9-141
//   AtmosphericScatterer                                src/Render/AtmosphericScatterer.cs:9-119
//   LoadScene()                                         src/MainWindow.cs:208-267
// (all relative to /root/reference/OpenTK-PathTracer/).  Same member names, argument meaning and error behaviour:
// where the reference throws a C# exception, these throw std::runtime_error carrying pt_last_error().
// Every method is a thin call into libmi355pt.so — there is no integrator code here.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mi355pt.h"

namespace opentk_pathtracer {

struct Vector3 {
    float X = 0, Y = 0, Z = 0;
    Vector3() = default;
    Vector3(float x, float y, float z) : X(x), Y(y), Z(z) {}
    explicit Vector3(float v) : X(v), Y(v), Z(v) {}
    Vector3 operator+(const Vector3 &o) const { return {X + o.X, Y + o.Y, Z + o.Z}; }
    Vector3 operator-(const Vector3 &o) const { return {X - o.X, Y - o.Y, Z - o.Z}; }
    Vector3 operator*(float s) const { return {X * s, Y * s, Z * s}; }
    Vector3 operator/(float s) const { return {X / s, Y / s, Z / s}; }
    static float Dot(const Vector3 &a, const Vector3 &b) { return a.X * b.X + a.Y * b.Y + a.Z * b.Z; }
    static Vector3 Cross(const Vector3 &a, const Vector3 &b)
    {
        return {a.Y * b.Z - a.Z * b.Y, a.Z * b.X - a.X * b.Z, a.X * b.Y - a.Y * b.X};
    }
    Vector3 Normalized() const { return *this / std::sqrt(Dot(*this, *this)); }
};
struct Vector4 {
    float X = 0, Y = 0, Z = 0, W = 0;
    static constexpr int SizeInBytes = 16;
};

// OpenTK 3.3.2 Matrix4 (row-major, row vectors) — published formulas of the NuGet package pinned in
// OpenTK-PathTracer.csproj:42; not vendored in the reference tree.
struct Matrix4 {
    float M[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    static Matrix4 LookAt(const Vector3 &eye, const Vector3 &target, const Vector3 &up)
    {
        Vector3 z = (eye - target).Normalized(), x = Vector3::Cross(up, z).Normalized(), y = Vector3::Cross(z, x).Normalized();
        Matrix4 r;
        float rows[4][4] = {{x.X, y.X, z.X, 0}, {x.Y, y.Y, z.Y, 0}, {x.Z, y.Z, z.Z, 0},
                            {-Vector3::Dot(x, eye), -Vector3::Dot(y, eye), -Vector3::Dot(z, eye), 1}};
        std::memcpy(r.M, rows, sizeof rows);
        return r;
    }
    static Matrix4 CreatePerspectiveFieldOfView(float fovy, float aspect, float zNear, float zFar)
    {
        float yMax = zNear * (float)std::tan(0.5f * fovy), yMin = -yMax, xMin = yMin * aspect, xMax = yMax * aspect;
        Matrix4 r;
        float rows[4][4] = {{2.0f * zNear / (xMax - xMin), 0, 0, 0},
                            {0, 2.0f * zNear / (yMax - yMin), 0, 0},
                            {(xMax + xMin) / (xMax - xMin), (yMax + yMin) / (yMax - yMin), -(zFar + zNear) / (zFar - zNear), -1},
                            {0, 0, -(2.0f * zFar * zNear) / (zFar - zNear), 0}};
        std::memcpy(r.M, rows, sizeof rows);
        return r;
    }
    Matrix4 Inverted() const // Gauss-Jordan with partial pivoting, evaluated in double and rounded once
    {
        double a[4][8];
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) { a[i][j] = M[i][j]; a[i][4 + j] = i == j; }
        for (int c = 0; c < 4; c++) {
            int p = c;
            for (int r = c + 1; r < 4; r++) if (std::fabs(a[r][c]) > std::fabs(a[p][c])) p = r;
            if (a[p][c] == 0.0) throw std::runtime_error("Matrix is singular and cannot be inverted.");
            for (int j = 0; j < 8; j++) std::swap(a[c][j], a[p][j]);
            double d = a[c][c];
            for (int j = 0; j < 8; j++) a[c][j] /= d;
            for (int r = 0; r < 4; r++)
                if (r != c) { double f = a[r][c]; for (int j = 0; j < 8; j++) a[r][j] -= f * a[c][j]; }
        }
        Matrix4 r;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.M[i][j] = (float)a[i][4 + j];
        return r;
    }
};
inline float DegreesToRadians(float deg) { return deg * (float)(3.14159265358979323846 / 180.0); } // MathHelper

class NativeError : public std::runtime_error {
public:
    int code;
    NativeError(int c, const std::string &m) : std::runtime_error("libmi355pt error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void Check(int rc, pt_handle h)
{
    if (rc != PT_OK) throw NativeError(rc, pt_last_error(h));
}

// ---- BufferObject as used for UBO 0 / UBO 1: only SubData matters on this path
class UniformBuffer {
public:
    enum Kind { BasicData, GameObjects };
    UniformBuffer(pt_handle h, Kind k) : h_(h), kind_(k) {}
    int Size() const { return kind_ == BasicData ? PT_BASIC_DATA_UBO_SIZE : PT_GAME_OBJECTS_UBO_SIZE; }
    void SubData(int offset, int size, const void *data) const
    {
        Check(kind_ == BasicData ? pt_upload_basic_data(h_, offset, size, data) : pt_upload_game_objects(h_, offset, size, data), h_);
    }
private:
    pt_handle h_;
    Kind kind_;
};

// ---- BaseSTD140Compatible (src/BaseSTD140Compatible.cs:6-17)
class BaseSTD140Compatible {
public:
    virtual ~BaseSTD140Compatible() = default;
    virtual int BufferOffset() const = 0;
    virtual std::vector<Vector4> GetGPUFriendlyData() const = 0;
    void Upload(const UniformBuffer &buffer) const
    {
        std::vector<Vector4> data = GetGPUFriendlyData();
        buffer.SubData(BufferOffset(), Vector4::SizeInBytes * (int)data.size(), data.data());
    }
};

// ---- Material (src/Material.cs:7-61)
class Material : public BaseSTD140Compatible {
public:
    static constexpr int GPU_INSTANCE_SIZE = 16 * 4;
    Vector3 Albedo{1, 1, 1}, Emissiv, AbsorbanceColor;
    float SpecularChance = 0, SpecularRoughness = 0, IOR = 1, RefractionChance = 0, RefractionRoughnes = 0;
    Material() = default;
    Material(Vector3 albedo, Vector3 emissiv, Vector3 refractionColor, float specularChance, float specularRoughness,
             float indexOfRefraction, float refractionChance, float refractionRoughnes)
        : Albedo(albedo), Emissiv(emissiv), AbsorbanceColor(refractionColor),
          SpecularChance(std::fmin(std::fmax(specularChance, 0.0f), 1.0f)), SpecularRoughness(specularRoughness),
          IOR(std::fmax(indexOfRefraction, 1.0f)), RefractionRoughnes(refractionRoughnes)
    {
        RefractionChance = std::fmin(std::fmax(refractionChance, 0.0f), 1.0f - SpecularChance); // Material.cs:29
    }
    static Material Zero() { return Material(Vector3(1.0f), Vector3(0.0f), Vector3(0.0f), 0, 0, 1, 0, 0); }
    int BufferOffset() const override { throw std::logic_error("Material is not meant to be directly uploaded to the GPU"); }
    std::vector<Vector4> GetGPUFriendlyData() const override // Material.cs:35-51
    {
        return {{Albedo.X, Albedo.Y, Albedo.Z, SpecularChance},
                {Emissiv.X, Emissiv.Y, Emissiv.Z, SpecularRoughness},
                {AbsorbanceColor.X, AbsorbanceColor.Y, AbsorbanceColor.Z, RefractionChance},
                {RefractionRoughnes, IOR, 0, 0}};
    }
};

class BaseGameObject : public BaseSTD140Compatible {
public:
    opentk_pathtracer::Material Material;
    Vector3 Position;
};

constexpr int MAX_GAMEOBJECTS_SPHERES = 256, MAX_GAMEOBJECTS_CUBOIDS = 64; // MainWindow.cs:17

// ---- Sphere (src/GameObjects/Sphere.cs:6-51)
class Sphere : public BaseGameObject {
public:
    static constexpr int GPU_INSTANCE_SIZE = 16 + opentk_pathtracer::Material::GPU_INSTANCE_SIZE;
    int Instance;
    float Radius;
    Sphere(Vector3 position, float radius, int instance, const opentk_pathtracer::Material &material)
        : Instance(instance), Radius(radius) { Position = position; Material = material; }
    int BufferOffset() const override { return 0 + Instance * GPU_INSTANCE_SIZE; }
    std::vector<Vector4> GetGPUFriendlyData() const override
    {
        std::vector<Vector4> d{{Position.X, Position.Y, Position.Z, Radius}};
        for (const Vector4 &v : Material.GetGPUFriendlyData()) d.push_back(v);
        return d;
    }
};

// ---- Cuboid (src/GameObjects/Cuboid.cs:6-53)
class Cuboid : public BaseGameObject {
public:
    static constexpr int GPU_INSTANCE_SIZE = 16 * 2 + opentk_pathtracer::Material::GPU_INSTANCE_SIZE;
    int Instance;
    Vector3 Dimensions;
    Cuboid(Vector3 position, Vector3 dimensions, int instance, const opentk_pathtracer::Material &material)
        : Instance(instance), Dimensions(dimensions) { Position = position; Material = material; }
    int BufferOffset() const override { return Sphere::GPU_INSTANCE_SIZE * MAX_GAMEOBJECTS_SPHERES + Instance * GPU_INSTANCE_SIZE; }
    Vector3 Min() const { return Position - Dimensions * 0.5f; }
    Vector3 Max() const { return Position + Dimensions * 0.5f; }
    std::vector<Vector4> GetGPUFriendlyData() const override
    {
        Vector3 mn = Min(), mx = Max();
        std::vector<Vector4> d{{mn.X, mn.Y, mn.Z, 0}, {mx.X, mx.Y, mx.Z, 0}};
        for (const Vector4 &v : Material.GetGPUFriendlyData()) d.push_back(v);
        return d;
    }
};

// ---- Camera (src/Camera.cs:7-83): only the pose matters to the integrator
class Camera {
public:
    Vector3 Position, ViewDir, Up;
    float LookX, LookY;
    Matrix4 View;
    Camera(Vector3 position, Vector3 up, float lookX = -90.0f, float lookY = 0.0f) : Position(position), Up(up), LookX(lookX), LookY(lookY)
    {
        ViewDir.X = std::cos(DegreesToRadians(LookX)) * std::cos(DegreesToRadians(LookY));
        ViewDir.Y = std::sin(DegreesToRadians(LookY));
        ViewDir.Z = std::sin(DegreesToRadians(LookX)) * std::cos(DegreesToRadians(LookY));
        View = GenerateMatrix(Position, ViewDir, Up);
    }
    static Matrix4 GenerateMatrix(Vector3 position, Vector3 viewDir, Vector3 up) { return Matrix4::LookAt(position, position + viewDir, up); }
};

class PathTracer;

// ---- EnvironmentMap: what a cube Texture is on this path (6 faces, RGBA32F or SRGB8_A8)
struct EnvironmentMap {
    int Size = 0, Format = PT_ENV_RGBA32F;
    std::vector<uint8_t> Faces[6];
    const void *FacePtr(int f) const { return Faces[f].data(); }
};

// ---- PathTracer (src/Render/PathTracer.cs:9-141)
class PathTracer {
public:
    PathTracer(const EnvironmentMap *environmentMap, int width, int height, int rayDepth, int spp, float focalLength,
               float apertureDiamater, int device = 0)
        : rayDepth_(rayDepth), spp_(spp), focalLength_(focalLength), apertureDiameter_(apertureDiamater), width_(width), height_(height)
    {
        int rc = pt_create(device, width, height, &h_);
        if (rc != PT_OK) throw NativeError(rc, pt_last_error(nullptr));
        PushParams();
        if (environmentMap) SetEnvironmentMap(*environmentMap);
    }
    // The same renderer row-tiled over several GPUs of this process (pt_create_multi; no reference counterpart): every
    // member below works unchanged, Result() / Present() gather over xGMI inside the library.
    PathTracer(const EnvironmentMap *environmentMap, const std::vector<int> &devices, int width, int height, int rayDepth, int spp,
               float focalLength, float apertureDiamater)
        : rayDepth_(rayDepth), spp_(spp), focalLength_(focalLength), apertureDiameter_(apertureDiamater), width_(width), height_(height)
    {
        int rc = pt_create_multi(devices.data(), (int)devices.size(), width, height, &h_);
        if (rc != PT_OK) throw NativeError(rc, pt_last_error(nullptr));
        PushParams();
        if (environmentMap) SetEnvironmentMap(*environmentMap);
    }
    ~PathTracer() { if (h_) pt_destroy(h_); }
    PathTracer(const PathTracer &) = delete;
    PathTracer &operator=(const PathTracer &) = delete;

    // the six uniform-setting properties, PathTracer.cs:11-83
    int NumSpheres() const { return numSpheres_; }
    void NumSpheres(int v) { numSpheres_ = v; PushParams(); }
    int NumCuboids() const { return numCuboids_; }
    void NumCuboids(int v) { numCuboids_ = v; PushParams(); }
    int RayDepth() const { return rayDepth_; }
    void RayDepth(int v) { rayDepth_ = v; PushParams(); }
    int SPP() const { return spp_; }
    void SPP(int v) { spp_ = v; PushParams(); }
    float FocalLength() const { return focalLength_; }
    void FocalLength(float v) { focalLength_ = v; PushParams(); }
    float ApertureDiameter() const { return apertureDiameter_; }
    void ApertureDiameter(float v) { apertureDiameter_ = v; PushParams(); }

    void SetEnvironmentMap(const EnvironmentMap &env) // `EnvironmentMap = ...`, PathTracer.cs:85
    {
        const void *faces[6];
        for (int f = 0; f < 6; f++) faces[f] = env.FacePtr(f);
        Check(pt_set_environment(h_, env.Size, env.Format, faces), h_);
    }
    int Samples() const { int f = 0; Check(pt_get_frame_index(h_, &f), h_); return f * spp_; } // PathTracer.cs:112
    void Render() { Check(pt_render(h_, nullptr), h_); }                                        // PathTracer.cs:114-129
    void SetFrameBatch(int maxFrames) { Check(pt_set_frame_batch(h_, maxFrames), h_); }         // frames one launch may pipeline
    void SetSize(int width, int height) { Check(pt_set_size(h_, width, height), h_); width_ = width; height_ = height; } // :131-135
    void ResetRenderer() { Check(pt_reset(h_), h_); }                                           // :137-140

    // `Result` (the RGBA32F texture ScreenEffect samples, MainWindow.cs:51), read back to the host
    std::vector<float> Result() const
    {
        std::vector<float> img((size_t)width_ * height_ * 4);
        Check(pt_read_result(h_, img.data(), 0), h_);
        return img;
    }
    // ScreenEffect.Render(PathTracer.Result) — src/Render/ScreenEffect.cs:29-37 — tone-mapped RGBA8 image
    std::vector<uint8_t> Present() const
    {
        std::vector<uint8_t> img((size_t)width_ * height_ * 4);
        Check(pt_present_rgba8(h_, img.data(), 0), h_);
        return img;
    }
    // Accumulation checkpoint (same file as opentk-pathtracer_amd/checkpoint.py): 8-byte magic, int32 x 8
    // (width, height, y0, rows, band rows / world / rank, frame index), int32 x 2 (depth, spp), float x 2 (focal length,
    // aperture), then the raw RGBA32F rows.  The C++ mirror renders whole images (no tiling).
    // Non-blocking present for the frame loop (MainWindow.cs:49-56): PresentAsync snapshots the frames rendered so far into
    // the library's pinned image of `slot` (0..PT_PRESENT_SLOTS-1) while later Render() calls proceed; PresentWait returns that
    // image (width * height RGBA8, row 0 = bottom; valid until the slot is presented into again) and the frame index it shows.
    void PresentAsync(int slot) { Check(pt_present_rgba8_async(h_, slot), h_); }
    const uint8_t *PresentWait(int slot, int *frameIndex = nullptr)
    {
        const uint8_t *img = nullptr;
        size_t pitch = 0;
        Check(pt_present_wait(h_, slot, &img, &pitch, frameIndex), h_);
        return img;
    }

    void SaveCheckpoint(const std::string &path) const
    {
        int frame = 0;
        Check(pt_get_frame_index(h_, &frame), h_);
        std::vector<float> img = Result();
        std::FILE *f = std::fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot open " + path);
        const int32_t ints[10] = {width_, height_, 0, height_, 0, 1, 0, frame, rayDepth_, spp_};
        const float lens[2] = {focalLength_, apertureDiameter_};
        bool ok = std::fwrite("PTCKPT1\0", 1, 8, f) == 8 && std::fwrite(ints, 4, 10, f) == 10 && std::fwrite(lens, 4, 2, f) == 2 &&
                  std::fwrite(img.data(), 4, img.size(), f) == img.size();
        ok = (std::fclose(f) == 0) && ok;
        if (!ok) throw std::runtime_error("short write to " + path);
    }
    int LoadCheckpoint(const std::string &path) // returns the restored frame index
    {
        std::FILE *f = std::fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open " + path);
        char magic[8];
        int32_t ints[10];
        float lens[2];
        std::vector<float> img((size_t)width_ * height_ * 4);
        bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "PTCKPT1\0", 8) == 0 && std::fread(ints, 4, 10, f) == 10 &&
                  std::fread(lens, 4, 2, f) == 2;
        // this class renders whole images: a checkpoint of one rank's rows (contiguous block or block-cyclic bands,
        // ints[4..6] = band rows / world / rank) is not a checkpoint of this renderer
        ok = ok && ints[0] == width_ && ints[1] == height_ && ints[2] == 0 && ints[3] == height_ && ints[4] == 0 && ints[7] >= 0 &&
             ints[8] == rayDepth_ && ints[9] == spp_ && lens[0] == focalLength_ && lens[1] == apertureDiameter_;
        ok = ok && std::fread(img.data(), 4, img.size(), f) == img.size() && std::fgetc(f) == EOF;
        std::fclose(f);
        if (!ok) throw std::runtime_error(path + ": not a checkpoint of this renderer (size, depth, spp or lens differ, or corrupt)");
        Check(pt_write_result(h_, img.data(), 0, ints[7]), h_);
        return ints[7];
    }
    // The GUI's screenshot (Gui.cs:28-33 -> Framebuffer.cs:67-82), headless: the displayed image, flipped vertically like the
    // reference's, as a binary PPM (no PNG encoder in the C++ standard library; checkpoint.py writes PNG).
    void SaveScreenshotPPM(const std::string &path) const
    {
        std::vector<uint8_t> img = Present();
        std::FILE *f = std::fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot open " + path);
        std::fprintf(f, "P6\n%d %d\n255\n", width_, height_);
        for (int y = height_ - 1; y >= 0; y--)
            for (int x = 0; x < width_; x++) std::fwrite(&img[((size_t)y * width_ + x) * 4], 1, 3, f);
        std::fclose(f);
    }
    int Width() const { return width_; }
    int Height() const { return height_; }
    pt_handle Handle() const { return h_; }
    UniformBuffer BasicDataUBO() const { return UniformBuffer(h_, UniformBuffer::BasicData); }     // MainWindow.cs:195-197
    UniformBuffer GameObjectsUBO() const { return UniformBuffer(h_, UniformBuffer::GameObjects); } // MainWindow.cs:199-201

private:
    void PushParams() { Check(pt_set_params(h_, numSpheres_, numCuboids_, rayDepth_, spp_, focalLength_, apertureDiameter_), h_); }
    pt_handle h_ = nullptr;
    int numSpheres_ = 0, numCuboids_ = 0, rayDepth_, spp_;
    float focalLength_, apertureDiameter_;
    int width_, height_;
};

// ---- AtmosphericScatterer (src/Render/AtmosphericScatterer.cs:9-119); its Result becomes the tracer's environment
class AtmosphericScatterer {
public:
    int ISteps = 50, JSteps = 15; // :92-93
    float Time = 0.5f, LightIntensity = 15.0f; // :91,94
    AtmosphericScatterer(PathTracer &tracer, int size) : tracer_(tracer), size_(size)
    {
        Matrix4 invProjection = Matrix4::CreatePerspectiveFieldOfView(DegreesToRadians(90.0f), 1, 0.1f, 10.0f).Inverted();
        const Vector3 dirs[6] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
        const Vector3 ups[6] = {{0, -1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}, {0, -1, 0}, {0, -1, 0}};
        Check(pt_atmosphere_upload_data(tracer_.Handle(), 0, 64, invProjection.M), tracer_.Handle());
        for (int i = 0; i < 6; i++) {
            Matrix4 inv = Camera::GenerateMatrix(Vector3(0.0f), dirs[i], ups[i]).Inverted();
            Check(pt_atmosphere_upload_data(tracer_.Handle(), 64 + 64 * i, 64, inv.M), tracer_.Handle());
        }
    }
    void SetSize(int size) { size_ = size; } // :115-118
    void Render()                            // :102-113; selects the cube as the tracer's EnvironmentMap (MainWindow.cs:189)
    {
        float a = DegreesToRadians(Time * 360.0f);
        float lightPos[3] = {0.0f * 149600000e3f, std::sin(a) * 149600000e3f, std::cos(a) * 149600000e3f}; // :41
        Check(pt_atmosphere_render(tracer_.Handle(), size_, ISteps, JSteps, lightPos, std::fmax(LightIntensity, 0.0f)), tracer_.Handle());
    }
private:
    PathTracer &tracer_;
    int size_;
};

// ---- MainWindow's scene + camera upload logic (src/MainWindow.cs:131-132,208-267,278-279)
struct Scene {
    std::vector<std::unique_ptr<BaseGameObject>> GameObjects;
    int NumSpheres = 0, NumCuboids = 0;
};

// The data of LoadScene() (MainWindow.cs:208-267): 48 spheres + 7 cuboids, no device access
inline Scene BuildDefaultScene()
{
    constexpr float EPSILON = 0.005f; // MainWindow.cs:18
    Scene sc;
    float width = 40.0f, height = 25.0f, depth = 25.0f;
    int balls = 6;
    float radius = 1.3f;
    Vector3 dimensions(width * 0.6f, height, depth);
    for (float x = 0; x < balls; x++)
        for (float y = 0; y < balls; y++)
            sc.GameObjects.emplace_back(new Sphere(
                Vector3(dimensions.X / balls * x * 1.1f - dimensions.X / 2, (dimensions.Y / balls) * y - dimensions.Y / 2 + radius, -5),
                radius, sc.NumSpheres++,
                Material(Vector3(0.59f, 0.59f, 0.99f), Vector3(0), Vector3(0), x / (balls - 1), y / (balls - 1), 1.0f, 0.0f, 0.1f)));
    Vector3 delta = dimensions / (float)balls;
    for (float x = 0; x < balls; x++) {
        Material material = Material::Zero();
        material.Albedo = Vector3(0.9f, 0.25f, 0.25f);
        material.SpecularChance = 0.02f;
        material.IOR = 1.05f;
        material.RefractionChance = 0.98f;
        material.AbsorbanceColor = Vector3(1, 2, 3) * (x / balls);
        sc.GameObjects.emplace_back(new Sphere(Vector3(-dimensions.X / 2 + radius + delta.X * x, 3.0f, -20.0f), radius, sc.NumSpheres++, material));
        Material material1 = Material::Zero();
        material1.SpecularChance = 0.02f;
        material1.SpecularRoughness = x / balls;
        material1.IOR = 1.1f;
        material1.RefractionChance = 0.98f;
        material1.RefractionRoughnes = x / balls;
        material1.AbsorbanceColor = Vector3(0.0f);
        sc.GameObjects.emplace_back(new Sphere(Vector3(-dimensions.X / 2 + radius + delta.X * x, -6.0f, -20.0f), radius, sc.NumSpheres++, material1));
    }
    auto cuboid = [&](Vector3 pos, Vector3 dim, const Material &m) {
        Cuboid *c = new Cuboid(pos, dim, sc.NumCuboids++, m);
        sc.GameObjects.emplace_back(c);
        return c;
    };
    Cuboid *down = cuboid(Vector3(0.0f, -height / 2.0f, -10.0f), Vector3(width, EPSILON, depth),
                          Material(Vector3(0.2f, 0.04f, 0.04f), Vector3(0.0f), Vector3(0), 0.0f, 0.051f, 1.0f, 0.0f, 0.0f));
    Vector3 dp = down->Position, dd = down->Dimensions;
    cuboid(Vector3(0.0f, 18.495f - EPSILON, -4.0f), Vector3(dd.X * 0.3f, EPSILON, dd.Z * 0.3f),
           Material(Vector3(0.04f), Vector3(0.917f, 0.945f, 0.513f) * 5.0f, Vector3(0), 0.0f, 0.0f, 1.0f, 0.0f, 0.0f));
    cuboid(Vector3(dp.X, dp.Y + height / 2, dp.Z + depth / 2 - 5.0f), Vector3(width, height, EPSILON),
           Material(Vector3(0.37109375f, 0.67578125f, 0.3359375f), Vector3(0.0f), Vector3(0), 0.0f, 0.0f, 1.0f, 0.0f, 0.0f));
    cuboid(Vector3(dp.X, dp.Y + height / 2 + EPSILON, dp.Z - depth / 2), Vector3(width, height - EPSILON * 2, 0.3f),
           Material(Vector3(1.0f), Vector3(0), Vector3(0.01f), 0.04f, 0.0f, 1.0f, 0.954f, 0.0f));
    cuboid(Vector3(dp.X + width / 2, dp.Y + height / 2.0f, dp.Z), Vector3(EPSILON, height, depth),
           Material(Vector3(0.9453125f, 0.75390625f, 0.3046875f), Vector3(0.0f), Vector3(0), 1.0f, 0.19f, 1.0f, 0.0f, 0.0f));
    cuboid(Vector3(dp.X - width / 2, dp.Y + height / 2.0f, dp.Z), Vector3(EPSILON, height, depth),
           Material(Vector3(0.074219f, 0.25f, 0.453125f), Vector3(0.0f), Vector3(0), 0.0f, 0.0f, 1.0f, 0.0f, 0.0f));
    cuboid(Vector3(-15.0f, -10.5f + EPSILON, -15.0f), Vector3(3.0f, 6.0f, 3.0f),
           Material(Vector3(1.0f), Vector3(0.0f), Vector3(0), 0.0f, 0.0f, 1.0f, 0.0f, 0.0f));
    return sc;
}

// LoadScene(): build + Upload() every object + publish the counts (MainWindow.cs:210-211,265-266)
inline Scene LoadScene(PathTracer &pathTracer)
{
    Scene sc = BuildDefaultScene();
    pathTracer.NumSpheres(sc.NumSpheres);
    pathTracer.NumCuboids(sc.NumCuboids);
    UniformBuffer ubo = pathTracer.GameObjectsUBO();
    for (const auto &o : sc.GameObjects) o->Upload(ubo);
    return sc;
}

// The 26,624-byte image of the GameObjectsUBO after all uploads (what the GL buffer would contain)
inline std::vector<uint8_t> GameObjectsUboImage(const Scene &sc)
{
    std::vector<uint8_t> blob(PT_GAME_OBJECTS_UBO_SIZE, 0);
    for (const auto &o : sc.GameObjects) {
        std::vector<Vector4> d = o->GetGPUFriendlyData();
        std::memcpy(blob.data() + o->BufferOffset(), d.data(), d.size() * sizeof(Vector4));
    }
    return blob;
}

// The 144-byte BasicDataUBO image (InvProjection, InvView, ViewPos)
inline std::vector<uint8_t> BasicDataUboImage(const Camera &camera, int width, int height, float fovDegrees = 103.0f,
                                              float zNear = 0.005f, float zFar = 1000.0f)
{
    std::vector<uint8_t> blob(PT_BASIC_DATA_UBO_SIZE, 0);
    Matrix4 invProj = Matrix4::CreatePerspectiveFieldOfView(DegreesToRadians(fovDegrees), width / (float)height, zNear, zFar).Inverted();
    Matrix4 invView = camera.View.Inverted();
    std::memcpy(blob.data(), invProj.M, 64);
    std::memcpy(blob.data() + 64, invView.M, 64);
    float pos[3] = {camera.Position.X, camera.Position.Y, camera.Position.Z};
    std::memcpy(blob.data() + 128, pos, 12);
    return blob;
}


// OnResize + OnUpdateFrame camera uploads (MainWindow.cs:131-132,278-279)
inline void UploadCamera(PathTracer &pathTracer, const Camera &camera, int width, int height, float fovDegrees = 103.0f,
                         float zNear = 0.005f, float zFar = 1000.0f)
{
    UniformBuffer ubo = pathTracer.BasicDataUBO();
    Matrix4 inverseProjection = Matrix4::CreatePerspectiveFieldOfView(DegreesToRadians(fovDegrees), width / (float)height, zNear, zFar).Inverted();
    ubo.SubData(0, Vector4::SizeInBytes * 4, inverseProjection.M);
    Matrix4 invView = camera.View.Inverted();
    ubo.SubData(Vector4::SizeInBytes * 4, Vector4::SizeInBytes * 4, invView.M);
    float pos[4] = {camera.Position.X, camera.Position.Y, camera.Position.Z, 0.0f};
    ubo.SubData(Vector4::SizeInBytes * 8, Vector4::SizeInBytes, pos);
}

// OnUpdateFrame alone (MainWindow.cs:131-132): the reference writes InvView and ViewPos on EVERY focused update, moved or not.  The
// library compares the bytes with what it holds: an unmoved camera costs two memcmp calls and changes nothing.
inline void UploadCameraPerFrame(PathTracer &pathTracer, const Camera &camera)
{
    UniformBuffer ubo = pathTracer.BasicDataUBO();
    Matrix4 invView = camera.View.Inverted();
    ubo.SubData(Vector4::SizeInBytes * 4, Vector4::SizeInBytes * 4, invView.M);
    float pos[4] = {camera.Position.X, camera.Position.Y, camera.Position.Z, 0.0f};
    ubo.SubData(Vector4::SizeInBytes * 8, Vector4::SizeInBytes, pos);
}

} // namespace opentk_pathtracer
