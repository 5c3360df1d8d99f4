// What does a PURE streaming kernel sustain on this MI355X at the read : write mixes of the HBM-bound layers?  (DESIGN graded the
// det heads / conv8_2 / conv1_1 against an asserted "1:1 copy reaches ~5.3 TB/s"; the guide measures 6.29 TB/s for a float4 copy.)  Drives the
// library's own probe kernels (v2x-sim_amd/csrc/calibrate.hip: 16 B per lane, R read + W write streams, each fully coalesced) over
// mix x cache policy x workgroups per CU x bytes per launch, best of 5 launches each.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hbm_mix_probe.hip -o /tmp/hbm_mix_probe && /tmp/hbm_mix_probe
#include "../v2x-sim_amd/csrc/calibrate.hip"

void v2x_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}

int main() {
    const size_t cap = (size_t)4 << 30;
    void *src, *dst;
    if (hipMalloc(&src, cap) != hipSuccess || hipMalloc(&dst, cap) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    (void)hipMemset(src, 0x5a, cap);
    (void)hipMemset(dst, 0, cap);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int mixes[6][2] = {{1, 1}, {1, 3}, {2, 1}, {4, 1}, {1, 0}, {0, 1}};
    const size_t totals[2] = {(size_t)1 << 30, (size_t)4 << 30};
    printf("%-6s %-3s %-7s %-9s %10s\n", "mix", "nt", "wg/CU", "bytes", "TB/s");
    for (int m = 0; m < 6; ++m)
        for (int nt = 0; nt < 2; ++nt)
            for (int wg = 2; wg <= 16; wg *= 2)
                for (int t = 0; t < 2; ++t) {
                    const int r = mixes[m][0], w = mixes[m][1];
                    const long long units = (long long)(totals[t] / (16 * (size_t)(r + w)));
                    float best = 1e30f;
                    for (int rep = 0; rep < 6; ++rep) {
                        (void)hipEventRecord(e0, 0);
                        if (v2x_calib_stream(src, dst, units, r, w, nt, wg, 0) != V2X_OK) return 2;
                        (void)hipEventRecord(e1, 0);
                        (void)hipEventSynchronize(e1);
                        float ms = 0;
                        (void)hipEventElapsedTime(&ms, e0, e1);
                        if (rep > 0 && ms < best) best = ms;
                    }
                    printf("%d:%d    %-3d %-7d %-9zu %10.3f\n", r, w, nt, wg, (size_t)units * 16 * (r + w), (double)units * 16.0 * (r + w) / (best * 1e-3) / 1e12);
                }
    // the MFMA loop of the same file: random / constant operands, both shapes, 40 launches back to back each (~0.2 s: past the boost window)
    float *scratch;
    unsigned long long *clocks;
    (void)hipMalloc(&scratch, 4096);
    (void)hipMalloc(&clocks, 16);
    for (int random = 1; random >= 0; --random)
        for (int shape32 = 0; shape32 < 2; ++shape32) {
            double fl = 0;
            for (int l = 0; l < 20; ++l) (void)v2x_calib_mfma(scratch, 20000, random ? 7u + l : 0u, shape32, (uint64_t *)clocks, &fl, 0);
            (void)hipEventRecord(e0, 0);
            for (int l = 0; l < 20; ++l) (void)v2x_calib_mfma(scratch, 20000, random ? 77u + l : 0u, shape32, (uint64_t *)clocks, &fl, 0);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c[2];
            (void)hipMemcpy(c, clocks, 16, hipMemcpyDeviceToHost);
            printf("mfma %s %s operands: %8.1f TFLOP/s, sustained shader clock %.0f MHz\n", shape32 ? "32x32x16" : "16x16x32", random ? "random  " : "constant",
                   fl * 20 / (ms * 1e-3) / 1e12, c[1] ? 100.0 * (double)c[0] / (double)c[1] : 0.0);
        }
    return 0;
}
