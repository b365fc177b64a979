// TEST INFRASTRUCTURE -- part of the oracle/_ref recipe (see README.md). Not linked into the product.
//
// The hit / miss stages of the reference's ray-tracing pipeline -- shaders/pathtrace.rchit, pathtrace.rmiss, pathtraceShadow.rmiss --
// compiled from the lexically rewritten sources, each in its own namespace (they all include globals.glsl; this file holds the three
// that need nothing else).  The any-hit stage includes layouts.glsl and random.glsl and lives in ref_rtx_rahit.cpp.
// Every stage is exported as a function over the payload bytes the trace call named.
#include <cstring>
#include "ref_driver.h"

namespace glslc {
thread_local GlobalInvocationID gl_LaunchIDEXT, gl_LaunchSizeEXT;
thread_local float  gl_HitTEXT;
thread_local int    gl_PrimitiveID, gl_InstanceID, gl_InstanceCustomIndexEXT;
thread_local mat4x3 gl_ObjectToWorldEXT, gl_WorldToObjectEXT;
thread_local bool   gl_IgnoreIntersection;
RtxTraceFn          g_rtxTrace = nullptr;

namespace rchit {
#include "pathtrace.rchit"
}
void ref_rtx_rchit(void* payload, size_t bytes, vec2 attribs)
{
  std::memcpy(&rchit::prd, payload, bytes < sizeof(rchit::prd) ? bytes : sizeof(rchit::prd));
  rchit::bary = attribs;
  rchit::shader_main();
  std::memcpy(payload, &rchit::prd, bytes < sizeof(rchit::prd) ? bytes : sizeof(rchit::prd));
}
}  // namespace glslc

// the two miss stages declare the same include guard (globals.glsl): undefine it between the namespaces so that each sees the types
#undef GLOBALS_GLSL
namespace glslc {
namespace rmiss {
#include "pathtrace.rmiss"
}
void ref_rtx_miss0(void*, size_t) { rmiss::shader_main(); }
}  // namespace glslc
#undef GLOBALS_GLSL
namespace glslc {
namespace smiss {
#include "pathtraceShadow.rmiss"
}
void ref_rtx_miss1(void* payload_, size_t bytes)
{
  std::memcpy(&smiss::payload, payload_, bytes < sizeof(smiss::payload) ? bytes : sizeof(smiss::payload));
  smiss::shader_main();
  std::memcpy(payload_, &smiss::payload, bytes < sizeof(smiss::payload) ? bytes : sizeof(smiss::payload));
}
}  // namespace glslc
