/* ref_codec.c — instantiates the reference's OWN image codec as a shared library for the tests.
 *
 * TEST INFRASTRUCTURE ONLY.  The reference decodes its inputs with src/libs/stb_image.h (stbi_load(…, STBI_rgb_alpha),
 * reference src/lfLoader.cpp:2-3, 36) and writes its views with src/libs/stb_image_write.h (stbi_write_png, reference
 * src/interpolator.cu:10-11, 313).  Both are single-header C libraries vendored in the reference tree, so — unlike the CUDA
 * kernels — this part of the reference builds here as it lies: oracle/Makefile compiles THIS file with
 * -I$(REF)/src/libs into oracle/_ref/libref_codec.so.  Nothing of the reference is copied into this repository; the two
 * macros below are how the reference's own translation units instantiate the headers.
 * Used by tests/test_host_io.py to pin csrc/host/image_io.cpp (PNG / JPEG decode) to the reference's decoder, pixel for pixel.
 */
#define STB_IMAGE_IMPLEMENTATION
#include <stb_image.h>
#define STB_IMAGE_WRITE_IMPLEMENTATION
#include <stb_image_write.h>
