// Debug harness: run launch_preprocess on one synthetic frame with PA_DEBUG_DUMP and dump LDS of one sub-band.
#include "../../playaid_core_amd/csrc/pa_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace pa;
int main(int argc, char** argv) {
    const int H = 1080, W = 1920;
    std::vector<uint8_t> frame((size_t)H * W * 3);
    unsigned h = 12345;
    for (auto& v : frame) { h = h * 1664525u + 1013904223u; v = h >> 24; }
    double box[8] = {0.5, 0.5, 0.30, 0.20, 0.3, 0.3, 0.1641, 0.2917};
    uint8_t *dframe, *t1, *t2, *crops, *dbg; double* dbox; CropPlan* plans; int32_t *coef, *status;
    hipMalloc(&dframe, frame.size()); hipMemcpy(dframe, frame.data(), frame.size(), hipMemcpyHostToDevice);
    hipMalloc(&dbox, sizeof(box)); hipMemcpy(dbox, box, sizeof(box), hipMemcpyHostToDevice);
    hipMalloc(&plans, 2 * sizeof(CropPlan)); hipMalloc(&coef, 2 * 2 * 1920 * 17 * 4);
    hipMalloc(&t1, 2 * frame.size()); hipMalloc(&t2, 2 * frame.size()); hipMalloc(&crops, 2 * 128 * 128 * 3); hipMalloc(&status, 8);
    hipMalloc(&dbg, 1 << 20); hipMemset(dbg, 0, 1 << 20);
    PreprocParams p{}; p.frames = dframe; p.boxes = dbox; p.n_frames = 1; p.height = H; p.width = W; p.fighters = 2; p.padding = 30;
    p.swap_rb = 0; p.plans = plans; p.coef = coef; p.coef_dim = 1920; p.t1 = t1; p.t2 = t2; p.t_stride = frame.size();
    int32_t* fb; hipMalloc(&fb, 64); p.fallback_count = fb; p.fallback_list = fb + 4;
    p.crops_u8 = crops; p.crops_f32 = nullptr; p.status = status; p.dbg = dbg; p.dbg_crop = atoi(argv[1]); p.dbg_row = atoi(argv[2]);
    hipError_t e = launch_preprocess(p, 0); hipDeviceSynchronize();
    printf("launch: %s\n", hipGetErrorString(e));
    std::vector<uint8_t> hd(1 << 20); hipMemcpy(hd.data(), dbg, 1 << 20, hipMemcpyDeviceToHost);
    std::vector<uint8_t> hc(2 * 128 * 128 * 3); hipMemcpy(hc.data(), crops, hc.size(), hipMemcpyDeviceToHost);
    FILE* f = fopen("gpurun_out/dbg_fused.bin", "wb"); fwrite(hd.data(), 1, hd.size(), f); fclose(f);
    f = fopen("gpurun_out/dbg_frame.bin", "wb"); fwrite(frame.data(), 1, frame.size(), f); fclose(f);
    f = fopen("gpurun_out/dbg_crops.bin", "wb"); fwrite(hc.data(), 1, hc.size(), f); fclose(f);
    int* m = (int*)hd.data(); for (int i = 0; i < 24; ++i) printf("%d ", m[i]); printf("\n");
    return 0;
}
