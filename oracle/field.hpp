// ORACLE — test infrastructure only. Selects the configuration the oracle is compiled for:
//   default        GoldilocksBlake3Config (src/types.rs:24-29, 199-223)            -> libms_oracle.so
//   -DMSO_BABYBEAR BabyBearPoseidon2Config (src/test_circuits/baby_bear_config.rs) -> libms_oracle_bb.so
// The BabyBear build also renames the namespace (-Dmso=msob in the Makefile) so both libraries can sit in one process.
#pragma once
#ifdef MSO_BABYBEAR
#include "bb.hpp"
#else
#include "gl.hpp"
#endif
#include "ext.hpp"
