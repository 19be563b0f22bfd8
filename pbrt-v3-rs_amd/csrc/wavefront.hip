// placeholder: the wavefront renderer lands next
#include "scene_host.h"
namespace phost { void free_wavefront(PbrtHipScene*) {} }
extern "C" {
int pbrt_hip_render_path(PbrtHipScene* s, int, float, int, const int*, int, int, int, float*, float*, PbrtHipStats*) { return phost::set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "not yet"); }
int pbrt_hip_generate_camera_rays(PbrtHipScene* s, const int*, uint32_t, PbrtHipRay*, float*) { return phost::set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "not yet"); }
}
