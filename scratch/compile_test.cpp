#define private public
#include "../finmath-lib-cuda-extensions_amd/csrc/runtime.hpp"
#include "../finmath-lib-cuda-extensions_amd/csrc/kernels.h"
#include <cstdio>
using namespace fm;
int main() {
    Engine& e = Engine::get();
    // chunk of K components: inputs: L_0..L_{K-1}, dW, S_carry
    for (int K = 1; K <= 8; ++K) {
        for (int carry = 0; carry <= 1; ++carry) {
            std::vector<SsaOp> ops; std::vector<int> outs;
            const int n_in = K + 1 + carry; const int dW = K, Sc = K + 1;
            int S = carry ? Sc : -1;
            auto add = [&](int op, int a, int b, int c, double s) { ops.push_back({op, a, b, c, s}); return n_in + (int)ops.size() - 1; };
            for (int k = 0; k < K; ++k) {
                int t = add(FMHIP_OP_MULT_S, k, -1, -1, 0.5); t = add(FMHIP_OP_ADD_S, t, -1, -1, 1.0); int y = add(FMHIP_OP_VID_S, t, -1, -1, 0.01);
                S = (S < 0) ? y : add(FMHIP_OP_ADD, S, y, -1, 0);
                int drift = add(FMHIP_OP_MULT_S, S, -1, -1, 0.02);
                int t1 = add(FMHIP_OP_ADDPRODUCT_VS, k, drift, -1, 0.5);
                int ln = add(FMHIP_OP_ADDPRODUCT_VS, t1, dW, -1, 0.02);
                outs.push_back(ln);
            }
            outs.push_back(S);
            try { Program* p = e.compile(ops, n_in, outs, {}, nullptr, false); printf("K=%d carry=%d: ok variant %u uops %u\n", K, carry, p->proto.variant, p->proto.n_ops); }
            catch (const Error& er) { printf("K=%d carry=%d: FAIL %s\n", K, carry, er.what()); }
        }
    }
}
namespace fm {
hipError_t launch_program(const DevProgramArgs&, const uint64_t*, double*, uint32_t, uint32_t, hipStream_t) { return hipSuccess; }
hipError_t launch_finalize(const DevFinalizeArgs&, uint32_t, hipStream_t) { return hipSuccess; }
hipError_t launch_bm(const DevBmArgs&, uint32_t, hipStream_t) { return hipSuccess; }
hipError_t launch_fill(float*, float, int64_t, hipStream_t) { return hipSuccess; }
}
