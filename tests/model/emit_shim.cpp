// emit_shim.cpp -- TEST INFRASTRUCTURE ONLY: exposes the host emitter's number formatting
// (lz-ani_amd/host/emit.h) to the Python tests.
#include "../../lz-ani_amd/host/emit.h"
extern "C" int host_format_real(double v, int prec, char* out) { return (int)host::real_to_chars(v, out, prec); }
