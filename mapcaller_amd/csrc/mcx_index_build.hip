// mapcaller_amd/csrc/mcx_index_build.hip — GPU construction of the BWA-compatible index
// (replaces bwa_idx_build, reference src/BWT_Index/bwtindex.c:77-160).  Placeholder until the
// suffix-array builder lands: reports MCX_ERR_UNSUPPORTED rather than falling back to the CPU.
#include <hip/hip_runtime.h>
#include "../../include/mcx.h"

extern "C" int mcx_index_build(const char *, const char *, int)
{
    return MCX_ERR_UNSUPPORTED;
}
