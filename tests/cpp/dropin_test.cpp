// dropin_test.cpp -- the C++ mirror of RayMarchingCallback (csrc/host/renderer.hpp) end to end on
// the GPU, written the way main.rs:71-79 + renderer.rs:195-256 use the reference types.
// Prints a checksum line that tests/test_gpu_cpp_dropin.py compares with the oracle.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "host/renderer.hpp"
#include "host/scenes.hpp"

using namespace ray_marching;

int main(int argc, char** argv) {
    const uint32_t W = 96, H = 64;
    const char* out_path = argc > 1 ? argv[1] : nullptr;
    try {
        auto resources = RayMarchingResources::new_(0);
        auto controller = OrbitCameraController::new_({0, 0, 0}, 5.0f);
        controller.update(OrbitCameraControllerEvent::Orbit{{35.f, -25.f}});
        auto cb = RayMarchingCallback::new_(0.0f, scenes::g8(), {(float)W, (float)H}, controller.camera());
        cb.prepare(resources);
        std::vector<float> image((size_t)W * H * 4);
        cb.paint(resources, W, H, image.data());
        if (out_path) {
            FILE* f = std::fopen(out_path, "wb");
            if (!f) return 3;
            std::fwrite(image.data(), 4, image.size(), f);
            std::fclose(f);
        }
        // The same frame once more with the scene's specialised kernel (compiled by hipRTC in THIS process: the
        // system ROCm, no PyTorch anywhere), waiting for the compiler: must be the same bytes.
        resources.check(rm_set_option(resources.ctx(), RM_OPT_SPECIALIZE, 2));
        std::vector<float> again(image.size());
        cb.paint(resources, W, H, again.data());
        double specialised = 0.0, jit_ms = 0.0;
        resources.check(rm_get_info(resources.ctx(), RM_INFO_SPECIALIZED, &specialised));
        resources.check(rm_get_info(resources.ctx(), RM_INFO_JIT_COMPILE_MS, &jit_ms));
        if (std::memcmp(again.data(), image.data(), image.size() * 4) != 0) {
            std::fprintf(stderr, "specialised kernel and interpreter kernel disagree\n");
            return 4;
        }
        std::printf("specialised %d jit_ms %.0f\n", (int)specialised, jit_ms);
        // Extension: a tagged scene built with the C++ node types and a material table (argv[2] receives the frame).
        if (argc > 2) {
            const float table[6][3] = {{0.4f, 0.7f, 0.1f}, {0.9f, 0.15f, 0.1f}, {0.1f, 0.3f, 0.9f},
                                       {0.95f, 0.9f, 0.2f}, {0.8f, 0.8f, 0.8f}, {0.6f, 0.1f, 0.7f}};
            resources.check(rm_set_materials(resources.ctx(), 6, &table[0][0]));
            auto tagged = RayMarchingCallback::new_(0.0f, scenes::mat_mix(), {(float)W, (float)H}, controller.camera());
            tagged.prepare(resources);
            tagged.paint(resources, W, H, again.data());
            FILE* f = std::fopen(argv[2], "wb");
            if (!f) return 3;
            std::fwrite(again.data(), 4, again.size(), f);
            std::fclose(f);
        }
        // None scene: cmd_count = 0 (renderer.rs:224-227)
        auto none_cb = RayMarchingCallback::new_(0.0f, std::nullopt, {(float)W, (float)H}, controller.camera());
        none_cb.prepare(resources);
        none_cb.paint(resources, W, H, image.data());
        std::printf("dropin ok %ux%u\n", W, H);
    } catch (const RmException& e) {
        std::fprintf(stderr, "rm error %d: %s\n", e.status, e.what());
        return 2;
    }
    return 0;
}
