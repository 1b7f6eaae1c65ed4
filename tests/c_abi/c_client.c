/* A plain-C client of the drop-in boundary (include/pdfnet_hip.h): no Python, no torch -- device memory from the HIP runtime,
 * the library opened by name.  Runs y = relu(x w^T + b) (pdf_linear_fwd), its weight gradient (pdf_linear_bwd_weight) and
 * Adam's neighbour pdf_stream_wait on the GPU and checks them against loops on the host.
 * Build (tests/test_host_cpu.py compiles it on the CPU box, tests/test_ops_gpu.py runs it on the GPU box):
 *   gcc -O1 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tests/c_abi/c_client.c -o c_client \
 *       -L pdfnet_amd -lpdfnet_hip -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/pdfnet_amd -Wl,-rpath,/opt/rocm/lib */
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <hip/hip_runtime_api.h>
#include "pdfnet_hip.h"

#define CHECK(call) do { int rc_ = (int)(call); if (rc_ != 0) { fprintf(stderr, "%s -> %d (line %d)\n", #call, rc_, __LINE__); return 1; } } while (0)

static float rnd(unsigned* s) { *s = *s * 1664525u + 1013904223u; return ((*s >> 8) & 0xffff) / 65536.0f - 0.5f; }

int main(void) {
    const int M = 300, N = 48, K = 80;                   /* ragged M: the last tile is partial */
    float *x = malloc(sizeof(float) * M * K), *w = malloc(sizeof(float) * N * K), *b = malloc(sizeof(float) * N);
    float *y = malloc(sizeof(float) * M * N), *dw = malloc(sizeof(float) * N * K), *db = malloc(sizeof(float) * N);
    unsigned seed = 7;
    for (int i = 0; i < M * K; ++i) x[i] = rnd(&seed);
    for (int i = 0; i < N * K; ++i) w[i] = rnd(&seed);
    for (int i = 0; i < N; ++i) b[i] = rnd(&seed);
    float *dx, *dwt, *dbias, *dy, *ddw, *ddb, *ws;
    const long wsf = pdf_wgrad_workspace_floats(M, N, K);
    CHECK(hipMalloc((void**)&dx, sizeof(float) * M * K)); CHECK(hipMalloc((void**)&dwt, sizeof(float) * N * K));
    CHECK(hipMalloc((void**)&dbias, sizeof(float) * N)); CHECK(hipMalloc((void**)&dy, sizeof(float) * M * N));
    CHECK(hipMalloc((void**)&ddw, sizeof(float) * N * K)); CHECK(hipMalloc((void**)&ddb, sizeof(float) * N));
    CHECK(hipMalloc((void**)&ws, sizeof(float) * (wsf > 0 ? wsf : 1)));
    CHECK(hipMemcpy(dx, x, sizeof(float) * M * K, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dwt, w, sizeof(float) * N * K, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dbias, b, sizeof(float) * N, hipMemcpyHostToDevice));
    hipStream_t s0, s1;
    CHECK(hipStreamCreate(&s0)); CHECK(hipStreamCreate(&s1));
    CHECK(pdf_init());
    if (pdf_debug_callopts_size() != (int)sizeof(PdfCallOpts)) { fprintf(stderr, "PdfCallOpts layout mismatch\n"); return 3; }
    /* the explicit form: every option of the call in its argument list (here: none armed, statistics not requested) */
    PdfCallOpts opts;
    memset(&opts, 0, sizeof opts);
    CHECK(pdf_linear_fwd_x(dx, dwt, dbias, dy, M, N, K, K, K, N, PDF_ACT_RELU, s0, &opts));
    if (opts.stats_tiles != 0 || pdf_debug_armed_slots() != 0) { fprintf(stderr, "unexpected option state\n"); return 4; }
    CHECK(pdf_stream_wait(s1, s0));                      /* the weight gradient (of L = sum y, dy = 1[y > 0]... here dy := y) waits for y */
    CHECK(pdf_linear_bwd_weight(dx, dy, ddw, ddb, ws, wsf, M, N, K, K, N, 0, s1));
    CHECK(hipStreamSynchronize(s1));
    CHECK(hipMemcpy(y, dy, sizeof(float) * M * N, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(dw, ddw, sizeof(float) * N * K, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(db, ddb, sizeof(float) * N, hipMemcpyDeviceToHost));
    double ey = 0, ew = 0, eb = 0;
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double a = b[n];
            for (int k = 0; k < K; ++k) a += (double)x[m * K + k] * w[n * K + k];
            if (a < 0) a = 0;
            ey = fmax(ey, fabs(a - y[m * N + n]));
        }
    for (int n = 0; n < N; ++n) {
        double sb = 0;
        for (int m = 0; m < M; ++m) sb += y[m * N + n];
        eb = fmax(eb, fabs(sb - db[n]));
        for (int k = 0; k < K; ++k) {
            double a = 0;
            for (int m = 0; m < M; ++m) a += (double)y[m * N + n] * x[m * K + k];
            ew = fmax(ew, fabs(a - dw[n * K + k]));
        }
    }
    printf("c_client: max |err|  y %.2e  dW %.2e  db %.2e\n", ey, ew, eb);
    if (!(ey < 1e-5 && ew < 1e-4 && eb < 1e-4)) { fprintf(stderr, "c_client: MISMATCH\n"); return 2; }
    /* argument errors come back as codes, nothing is launched */
    if (pdf_cast_bf16(dx, dy, 3, s0) == 0) { fprintf(stderr, "c_client: n %% 4 != 0 was accepted\n"); return 3; }
    printf("c_client: ok\n");
    return 0;
}
