// Multi-pass spectral path for grids whose member field does not fit one CU's LDS (N >= 128).
#include "common.hpp"

namespace qgx {

int large_prepare(const SpecDev &) { return QGX_OK; }
int large_q_to_qh(qgx_model *, const double *, double2 *, hipStream_t) {
    set_error("large-grid spectral path not built yet");
    return QGX_ERR_INVALID;
}
int large_qh_to_q(qgx_model *, const double2 *, double *, hipStream_t) {
    set_error("large-grid spectral path not built yet");
    return QGX_ERR_INVALID;
}
int large_invert(qgx_model *, hipStream_t) {
    set_error("large-grid spectral path not built yet");
    return QGX_ERR_INVALID;
}
int large_step(qgx_model *, const StepArgs &, hipStream_t) {
    set_error("large-grid spectral path not built yet");
    return QGX_ERR_INVALID;
}

}  // namespace qgx
