// pk_tables.hip -- the table kernels of libpokerl_hip.so for ONE seat count (-DPK_SEATS=N): explicit instantiations of the
// templates in pk_kernels.hpp.  One object per seat count, compiled in parallel (pokerl_amd/build.py); pk_api.hip declares
// the same instantiations `extern template` and launches them.  gfx950 only.
#define PK_TABLES_ONLY
#include "pk_kernels.hpp"

#ifndef PK_SEATS
#error "compile with -DPK_SEATS=<number of seats>"
#endif
static_assert(PK_SEATS >= PK_MIN_PLAYERS && PK_SEATS <= PK_MAX_PLAYERS, "seat count outside the ABI's range");

PK_TABLE_KERNELS(PK_INSTANTIATE_KERNEL, PK_SEATS)
#if PK_SEATS <= 10
PK_TABLE_KERNELS_LE10(PK_INSTANTIATE_KERNEL, PK_SEATS)
#endif
#if PK_SEATS <= 6
PK_TABLE_KERNELS_LE6(PK_INSTANTIATE_KERNEL, PK_SEATS)
#endif
