// Self-checks of the host classes that need a device but no input files; exported from librte_rrtmgp_hip.so for the GPU
// tests (tests/test_gpu_host_classes.py). Each returns 0 on success and a non-zero code naming the failed check.
#include <cmath>
#include <vector>
#include "Array.h"
#include "Optical_props.h"

namespace
{
    Optical_props_gpu two_bands()
    {
        Array<Float,2> wvn({2, 2});
        wvn({1, 1}) = 10.; wvn({2, 1}) = 250.; wvn({1, 2}) = 250.; wvn({2, 2}) = 500.;
        Array<int,2> lims({2, 2});
        lims({1, 1}) = 1; lims({2, 1}) = 3; lims({1, 2}) = 4; lims({2, 2}) = 6;
        return Optical_props_gpu(wvn, lims);
    }
}

extern "C"
{
// Optical_props_2str_gpu::delta_scale() on gas-only optical properties whose asymmetry is held in the lazy "g == 0"
// state (what the SW gas optics leave behind): the result must be the identity on tau and ssa, g must read back as
// zeros, and the stale contents of the g array must never reach the kernel (reference semantics with g = 0:
// /root/reference/src_kernels_cuda/optical_props_kernels.cu:140-161).
int rrx_host_selftest_delta_scale_gzero(void)
{
    try
    {
        const int ncol = 5, nlay = 4;
        Optical_props_2str_gpu op(ncol, nlay, two_bands());
        const int n = ncol*nlay*op.get_ngpt();
        Array<Float,3> tau({ncol, nlay, op.get_ngpt()}), ssa({ncol, nlay, op.get_ngpt()});
        for (int i=0; i<n; ++i) { tau.ptr()[i] = Float(0.01)*(i+1); ssa.ptr()[i] = Float(1.)/(i+2); }
        op.get_tau().set_data(tau); op.get_ssa().set_data(ssa);
        op.get_g().fill(Float(0.7));        // stale values of an earlier use of the workspace (e.g. cloud-incremented g)
        op.set_g_zero();
        if (op.get_g_or_null() != nullptr) return 1;
        op.delta_scale();
        Array<Float,3> t2(op.get_tau()), w2(op.get_ssa());
        for (int i=0; i<n; ++i)
            if (t2.ptr()[i] != tau.ptr()[i] || w2.ptr()[i] != ssa.ptr()[i]) return 2;
        if (op.get_g_or_null() != nullptr) return 3;               // still lazy: nothing materialised, nothing read
        Array<Float,3> g2(op.get_g());                              // first real access fills zeros
        for (int i=0; i<n; ++i) if (g2.ptr()[i] != Float(0.)) return 4;
        // and with a materialised g the kernel runs: f = g^2, tau' = tau (1 - ssa f)
        op.get_g().fill(Float(0.5));
        op.delta_scale();
        Array<Float,3> t3(op.get_tau());
        for (int i=0; i<n; ++i)
        {
            const Float want = tau.ptr()[i] * (Float(1.) - ssa.ptr()[i]*Float(0.25));
            if (std::abs(t3.ptr()[i] - want) > Float(1e-6)*std::abs(want)) return 5;
        }
        return 0;
    }
    catch (const std::exception&) { return 100; }
}
}
