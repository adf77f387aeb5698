// Stand-in for the reference's generated cl_kernels.h (CMakeLists.txt:66-102 embeds kernel/*.cl as C strings).  No OpenCL
// source is shipped here: each constant is a tag OpenCL::compile() maps to a cl::KernelId.  compile() also recognises the
// reference's real source strings by their kernel entry name, so an integrator may keep the generated header.
#pragma once
static const char* const kernel_raw2quad_cl = "vp:raw2quad";
static const char* const kernel_resampling_cl = "vp:resampling";
static const char* const kernel_gradientDot_cl = "vp:gradient_dotproduct";
static const char* const kernel_satHorizontal_cl = "vp:sat_horizontal";
static const char* const kernel_satVertical_cl = "vp:sat_vertical";
static const char* const kernel_satBlobCenter_cl = "vp:circle";
static const char* const kernel_blobList_cl = "vp:blobList";
static const char* const kernel_blobScore_cl = "vp:blobScore";
static const char* const kernel_quad2rgba_cl = "vp:quad2rgba";
static const char* const kernel_quad2nv12_cl = "vp:quad2nv12";
static const char* const kernel_rgba2nv12_cl = "vp:rgb2nv12";
static const char* const kernel_f2nv12_cl = "vp:f2nv12";
