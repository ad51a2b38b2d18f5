// Stand-alone probe (no library code) of two hipEventQuery hazards on ROCm 7.2, looked for after the GPU suite's one flaky failure
// ("operation not permitted on an event last recorded in a capturing stream" from hipEventQuery on a staging-ring slot event, with no
// stream capture anywhere in the process):
//   A. an event that was NEVER recorded, created after heap churn (streams and events created and destroyed)
//   B. an event last recorded on a stream that has since been DESTROYED (hipEventQuery dereferences the event's stream: the disassembly
//      of libamdhip64.so.7.2 reads [event + 8] and compares [that + 0x298] with 1 = hipStreamCaptureStatusActive)
// Prints how many queries of each kind returned something other than hipSuccess / hipErrorNotReady.  Case B provokes a use-after-free
// INSIDE the runtime on purpose (it also WRITES through the stale pointer): run it on a scratch box, once.
// Result on the MI355X box (gpurun call r06_c29): A 0 of 25 600; B 1 x "... event last recorded in a capturing stream" + 15 x "operation not
// permitted when stream is capturing" of 25 600 -- the two texts the GPU suite had shown.
//   hipcc --offload-arch=gfx950 -O2 tools/event_capture_probe.hip -o tools/alt/event_capture_probe && tools/alt/event_capture_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <string>
#include <vector>

__global__ void spin(int* p, int n) { for (int i = 0; i < n; i++) atomicAdd(p, 1); }

static void churn(int rounds) {
    for (int r = 0; r < rounds; r++) {
        std::vector<hipStream_t> ss(8);
        std::vector<hipEvent_t> ee(64);
        for (auto& s : ss) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (auto& e : ee) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        for (size_t i = 0; i < ee.size(); i++) (void)hipEventRecord(ee[i], ss[i % ss.size()]);
        for (auto& s : ss) (void)hipStreamSynchronize(s);
        for (auto& e : ee) (void)hipEventDestroy(e);
        for (auto& s : ss) (void)hipStreamDestroy(s);
    }
}

int main() {
    int* d = nullptr;
    (void)hipMalloc(&d, 4);
    std::map<std::string, int> errA, errB;
    int nA = 0, nB = 0;
    for (int round = 0; round < 200; round++) {
        churn(4);
        // A: fresh events, never recorded
        std::vector<hipEvent_t> fresh(128);
        for (auto& e : fresh) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        for (auto& e : fresh) {
            const hipError_t q = hipEventQuery(e);
            nA++;
            if (q != hipSuccess && q != hipErrorNotReady) { errA[hipGetErrorString(q)]++; (void)hipGetLastError(); }
        }
        // B: events recorded on streams that are then destroyed
        std::vector<hipStream_t> ss(8);
        for (auto& s : ss) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (size_t i = 0; i < fresh.size(); i++) {
            hipLaunchKernelGGL(spin, dim3(1), dim3(1), 0, ss[i % ss.size()], d, 10);
            (void)hipEventRecord(fresh[i], ss[i % ss.size()]);
        }
        for (auto& s : ss) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
        churn(4);                                        // other objects take the destroyed streams' memory
        for (auto& e : fresh) {
            const hipError_t q = hipEventQuery(e);
            nB++;
            if (q != hipSuccess && q != hipErrorNotReady) { errB[hipGetErrorString(q)]++; (void)hipGetLastError(); }
        }
        for (auto& e : fresh) (void)hipEventDestroy(e);
    }
    printf("A (never recorded): %d queries", nA);
    for (auto& kv : errA) printf("; %d x \"%s\"", kv.second, kv.first.c_str());
    printf("\nB (stream destroyed after the record): %d queries", nB);
    for (auto& kv : errB) printf("; %d x \"%s\"", kv.second, kv.first.c_str());
    printf("\n");
    // is the process still healthy?
    const hipError_t m = hipMemset(d, 0, 4);
    printf("hipMemset afterwards: %s\n", hipGetErrorString(m));
    return 0;
}
