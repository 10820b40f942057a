// render.h -- the reference's GPU render operators on the HIP back-end.
//
// Same names, call order and argument meaning as
// /root/reference/source/include/render.h:7-21:
//     render::setup::context(); render::setup::shaders(); render::setup::buffers();
//     render::push();                  // scene::tree/lights/camera -> device
//     render::GPU(image);              // caller-owned width*height*4 bytes, RGBA8
//     render::destroy();
// All return void; a failure prints "[Error] ..." to stdout and execution goes on,
// as the reference does (render.cpp:417-425).  render::ok() / render::error() add
// what the reference lacks: a way to ask whether the last operator succeeded.
//
// render::CPU / render::OMP (the reference's CPU loops) are not part of this
// library: the CPU path is the external baseline, not a fallback.
#pragma once

#include "../../include/raymarch_hip.h"
#include "scene.h"

namespace constants {
// Run-time counterparts of /root/reference/source/include/constants.h:11-27,
// initialised to the reference's values; render::push() uploads them.
extern rm_params params;
}  // namespace constants

namespace render {

// GPU() page-locks the caller's image the first time it sees it (hipHostRegister) and keeps that registration: the image
// must stay allocated until destroy() or unpin().  Free or reallocate it only after one of the two.
void GPU(unsigned char *image);

namespace setup {
// Extension: the devices the frame is rendered on, before context().  One device (the default: device 0) is the
// reference's case; with more, context() creates one rank per entry and GPU() shards every frame over them by row
// bands and gathers it (rm_multi_*: direct stores into the first device's frame, or RCCL where the devices have no peer access).
void devices(const int *device_ids, unsigned count);
void context(void);
void shaders(void);
void buffers(void);
}  // namespace setup

void push(void);
void destroy(void);

// Extensions over the reference's interface.
void GPU(float *image);      // float RGBA frame, as the CPU operators' Image2D<float4>
bool ok(void);               // did the last operator succeed
const char *error(void);     // message of the last failure
rm_context *context(void);   // the underlying C-ABI context (null before setup::context); rank 0's of a sharded set-up
rm_multi *multi(void);       // the sharded set-up's object (null with one device)
void unpin(void);            // drops the page-lock GPU() holds on the caller's image (before the caller frees it)

}  // namespace render
