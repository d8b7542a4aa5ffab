// pt_host_demo — replays the reference host's start-up and frame loop over the C ABI, in C++:
//   MainWindow.OnLoad  (src/MainWindow.cs:146-206): AtmosphericScatterer(256).Render() -> PathTracer(env = atmosphere,
//                      W, H, rayDepth 13, spp 1, focalLength 20, aperture 0.14) -> UBOs -> LoadScene()
//   OnResize / OnUpdateFrame camera uploads (:131-132,278-279), then N x PathTracer.Render() (:49)
// and writes the RGBA32F `Result` (raw floats, row 0 = bottom) so it can be compared with the oracle.
//
//   pt_host_demo render <W> <H> <frames> <out.f32> [rayDepth] [atmosphereSize] [env-out.f32 (the cube as pt_read_environment returns it)]
//   pt_host_demo frame-loop <W> <H> <frames> <out.rgba8> <devices e.g. 0 or 0,0,0>
//                      the frame loop of MainWindow.OnRenderFrame (:40-69) with the NON-BLOCKING present on one GPU or a group of
//                      GPUs: Render(); PresentAsync(f % 3); PresentWait((f + 1) % 3); writes the last image shown + its frame index
//   pt_host_demo dump-scene <out.bin>            (no GPU needed: the 26,624-byte GameObjectsUBO image)
//   pt_host_demo dump-camera <W> <H> <out.bin>   (no GPU needed: the 144-byte BasicDataUBO image)
#include <cstdio>
#include <cstdlib>
#include <string>

#include "pt_host.hpp"

using namespace opentk_pathtracer;

static void write_file(const char *path, const void *data, size_t n)
{
    FILE *f = std::fopen(path, "wb");
    if (!f || std::fwrite(data, 1, n, f) != n) { std::perror(path); std::exit(2); }
    std::fclose(f);
}

int main(int argc, char **argv)
{
    try {
        std::string mode = argc > 1 ? argv[1] : "";
        Camera camera(Vector3(-17.14f, 3.53f, -8.62f), Vector3(0, 1, 0), -32.2f, 0.8f); // MainWindow.cs:36
        if (mode == "dump-scene" && argc == 3) {
            Scene sc = BuildDefaultScene();
            auto blob = GameObjectsUboImage(sc);
            write_file(argv[2], blob.data(), blob.size());
            std::printf("%d spheres, %d cuboids\n", sc.NumSpheres, sc.NumCuboids);
            return 0;
        }
        if (mode == "dump-camera" && argc == 5) {
            auto blob = BasicDataUboImage(camera, std::atoi(argv[2]), std::atoi(argv[3]));
            write_file(argv[4], blob.data(), blob.size());
            return 0;
        }
        if (mode == "render" && argc >= 6) {
            int W = std::atoi(argv[2]), H = std::atoi(argv[3]), frames = std::atoi(argv[4]);
            int rayDepth = argc > 6 ? std::atoi(argv[6]) : 13, atmo = argc > 7 ? std::atoi(argv[7]) : 256;
            PathTracer pathTracer(nullptr, W, H, rayDepth, 1, 20.0f, 0.14f);
            AtmosphericScatterer atmosphericScatterer(pathTracer, atmo);
            atmosphericScatterer.Render();
            LoadScene(pathTracer);
            UploadCamera(pathTracer, camera, W, H);
            for (int i = 0; i < frames; i++) {
                UploadCameraPerFrame(pathTracer, camera); // OnUpdateFrame, MainWindow.cs:131-132: every update, moved or not
                pathTracer.Render();                      // OnRenderFrame, :49
            }
            std::vector<float> img = pathTracer.Result();
            write_file(argv[5], img.data(), img.size() * sizeof(float));
            if (argc > 8) { // the environment the frames were rendered with, read back through the C ABI
                std::vector<float> cube((size_t)6 * atmo * atmo * 4);
                int face = 0;
                if (pt_read_environment(pathTracer.Handle(), cube.data(), &face) != PT_OK || face != atmo) throw std::runtime_error("pt_read_environment failed");
                write_file(argv[8], cube.data(), cube.size() * sizeof(float));
            }
            std::printf("rendered %dx%d, %d samples/pixel\n", W, H, pathTracer.Samples());
            return 0;
        }
        if (mode == "frame-loop" && argc == 7) {
            int W = std::atoi(argv[2]), H = std::atoi(argv[3]), frames = std::atoi(argv[4]);
            std::vector<int> devices;
            for (const char *p = argv[6]; *p;) {
                devices.push_back(std::atoi(p));
                while (*p && *p != ',') p++;
                if (*p == ',') p++;
            }
            PathTracer pathTracer(nullptr, devices, W, H, 13, 1, 20.0f, 0.14f);
            AtmosphericScatterer atmosphericScatterer(pathTracer, 64);
            atmosphericScatterer.Render();
            LoadScene(pathTracer);
            UploadCamera(pathTracer, camera, W, H);
            const uint8_t *shown = nullptr;
            int shownFrame = 0, presented = 0;
            for (int f = 0; f < frames; f++) {
                UploadCameraPerFrame(pathTracer, camera);  // OnUpdateFrame, MainWindow.cs:131-132
                pathTracer.Render();                       // PathTracer.Render(), MainWindow.cs:49
                pathTracer.PresentAsync(presented % 3);    // instead of PostProcesser.Render(PathTracer.Result), :51
                if (++presented >= 3) shown = pathTracer.PresentWait(presented % 3, &shownFrame); // the frame of two calls ago: landed
            }
            for (int k = 1; k <= 2 && presented >= k; k++) shown = pathTracer.PresentWait((presented + k) % 3, &shownFrame); // drain
            write_file(argv[5], shown, (size_t)W * H * 4);
            std::printf("frame loop over %d device(s): %d frames rendered, last image shown is frame %d\n", (int)devices.size(), frames, shownFrame);
            return 0;
        }
        if (mode == "resume" && argc == 9) {
            // render framesA, checkpoint, restore into a NEW renderer, render framesB more
            int W = std::atoi(argv[2]), H = std::atoi(argv[3]), framesA = std::atoi(argv[4]), framesB = std::atoi(argv[5]);
            auto start = [&](PathTracer &pt, AtmosphericScatterer &atmo) {
                atmo.Render();
                LoadScene(pt);
                UploadCamera(pt, camera, W, H);
            };
            {
                PathTracer first(nullptr, W, H, 13, 1, 20.0f, 0.14f);
                AtmosphericScatterer atmo(first, 256);
                start(first, atmo);
                for (int i = 0; i < framesA; i++) first.Render();
                first.SaveCheckpoint(argv[7]);
            }
            PathTracer second(nullptr, W, H, 13, 1, 20.0f, 0.14f);
            AtmosphericScatterer atmo(second, 256);
            start(second, atmo);
            int restored = second.LoadCheckpoint(argv[7]);
            for (int i = 0; i < framesB; i++) second.Render();
            std::vector<float> img = second.Result();
            write_file(argv[6], img.data(), img.size() * sizeof(float));
            second.SaveScreenshotPPM(argv[8]);
            std::printf("resumed at frame %d, now %d samples/pixel\n", restored, second.Samples());
            return 0;
        }
        std::fprintf(stderr, "usage: pt_host_demo render W H frames out.f32 [rayDepth] [atmoSize] | frame-loop W H frames out.rgba8 devices | resume W H framesA framesB out.f32 ckpt shot.ppm | "
                             "dump-scene out.bin | dump-camera W H out.bin\n");
        return 1;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "pt_host_demo: %s\n", e.what()); // Program.cs:15-25: catch-all prints the exception
        return 3;
    }
}
