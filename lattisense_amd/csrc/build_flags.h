// build_flags.h — every compile-time switch of the library, in one place.
//
// The product library is built with NONE of them (lattisense_amd/build.py: build_native passes no -D and ignores the
// environment).  A/B and diagnostic builds go through build_variant, get their own file name and SONAME, and report what they
// were built with through lsa_build_flags(); the Python binding refuses to load a library with a non-empty flag string as the
// product (lattisense_amd/_native.py), and bench.py prints the string in its JSON line.
//
// Switches that change RESULTS (wrong-result diagnostics: arithmetic or memory traffic removed to time the rest) must be
// accompanied by -DLSA_DIAG_BUILD, which build_variant adds for them and which renames the library
// (liblattisense_amd_diag_<name>.so): such an object can never be mistaken for the product.
#pragma once

#if (defined(LSA_NTT_DIAG_NO_TWIDDLE_LOADS) || defined(LSA_NTT_DIAG_COPY_ONLY) || defined(LSA_NTT_DIAG_COMPUTE_ONLY) || defined(LSA_NTT_DIAG_TW8) || \
     defined(LSA_KS_DIAG_NO_MATH) || defined(LSA_KS_DIAG_NO_STORE) || defined(LSA_BC_DIAG_NO_MATH) || defined(LSA_BC_DIAG_NO_STORE)) && \
    !defined(LSA_DIAG_BUILD)
#error "wrong-result diagnostic switches need -DLSA_DIAG_BUILD (use lattisense_amd.build --variant: it adds the flag and renames the library)"
#endif

#define LSA_BF_STR2(x) #x
#define LSA_BF_STR(x) LSA_BF_STR2(x)

// one token per switch that differs from the product configuration (value switches: only when the value is not the default)
#define LSA_BUILD_FLAGS_TEXT_1 ""
#ifdef LSA_DIAG_BUILD
#define LSA_BF_DIAG " LSA_DIAG_BUILD"
#else
#define LSA_BF_DIAG ""
#endif
#ifdef LSA_NTT_DIAG_NO_TWIDDLE_LOADS
#define LSA_BF_A " LSA_NTT_DIAG_NO_TWIDDLE_LOADS"
#else
#define LSA_BF_A ""
#endif
#ifdef LSA_NTT_DIAG_COPY_ONLY
#define LSA_BF_B " LSA_NTT_DIAG_COPY_ONLY"
#else
#define LSA_BF_B ""
#endif
#ifdef LSA_NTT_DIAG_COMPUTE_ONLY
#define LSA_BF_C " LSA_NTT_DIAG_COMPUTE_ONLY"
#else
#define LSA_BF_C ""
#endif
#ifdef LSA_KS_DIAG_NO_MATH
#define LSA_BF_D " LSA_KS_DIAG_NO_MATH"
#else
#define LSA_BF_D ""
#endif
#ifdef LSA_KS_DIAG_NO_STORE
#define LSA_BF_E " LSA_KS_DIAG_NO_STORE"
#else
#define LSA_BF_E ""
#endif
#ifdef LSA_BC_DIAG_NO_MATH
#define LSA_BF_F " LSA_BC_DIAG_NO_MATH"
#else
#define LSA_BF_F ""
#endif
#ifdef LSA_BC_DIAG_NO_STORE
#define LSA_BF_G " LSA_BC_DIAG_NO_STORE"
#else
#define LSA_BF_G ""
#endif
#ifdef LSA_NTT_DIAG_STAMPS
#define LSA_BF_H " LSA_NTT_DIAG_STAMPS"
#else
#define LSA_BF_H ""
#endif
#ifdef LSA_NTT_STAGGER
#define LSA_BF_I " LSA_NTT_STAGGER=" LSA_BF_STR(LSA_NTT_STAGGER)
#else
#define LSA_BF_I ""
#endif
#ifdef LSA_NTT_TW_NATURAL
#define LSA_BF_J " LSA_NTT_TW_NATURAL"
#else
#define LSA_BF_J ""
#endif
#ifdef LSA_NTT_EXACT_BFLY
#define LSA_BF_K " LSA_NTT_EXACT_BFLY"
#else
#define LSA_BF_K ""
#endif
#ifdef LSA_NTT_NO_NT
#define LSA_BF_L " LSA_NTT_NO_NT"
#else
#define LSA_BF_L ""
#endif
#ifdef LSA_MAC128_COMPARE
#define LSA_BF_M " LSA_MAC128_COMPARE"
#else
#define LSA_BF_M ""
#endif
#ifdef LSA_BC_CARRY_COMPARES
#define LSA_BF_N " LSA_BC_CARRY_COMPARES"
#else
#define LSA_BF_N ""
#endif
#if defined(LSA_NTT_THREADS) && LSA_NTT_THREADS != 256
#define LSA_BF_O " LSA_NTT_THREADS=" LSA_BF_STR(LSA_NTT_THREADS)
#else
#define LSA_BF_O ""
#endif
#if defined(LSA_NTT_MAX_RHO) && LSA_NTT_MAX_RHO != 4
#define LSA_BF_P " LSA_NTT_MAX_RHO=" LSA_BF_STR(LSA_NTT_MAX_RHO)
#else
#define LSA_BF_P ""
#endif
#if defined(LSA_NTT_WAVES) && LSA_NTT_WAVES != 4
#define LSA_BF_Q " LSA_NTT_WAVES=" LSA_BF_STR(LSA_NTT_WAVES)
#else
#define LSA_BF_Q ""
#endif
#if defined(LSA_NTT_WAVES_FUSED) && LSA_NTT_WAVES_FUSED != 3
#define LSA_BF_R " LSA_NTT_WAVES_FUSED=" LSA_BF_STR(LSA_NTT_WAVES_FUSED)
#else
#define LSA_BF_R ""
#endif
#if defined(LSA_NTT_TAU) && LSA_NTT_TAU != 12
#define LSA_BF_S " LSA_NTT_TAU=" LSA_BF_STR(LSA_NTT_TAU)
#else
#define LSA_BF_S ""
#endif
#if defined(LSA_NTT_STORE_CHUNK) && LSA_NTT_STORE_CHUNK != 4
#define LSA_BF_T " LSA_NTT_STORE_CHUNK=" LSA_BF_STR(LSA_NTT_STORE_CHUNK)
#else
#define LSA_BF_T ""
#endif
#if defined(LSA_NTT_HEAD_ROUNDS) && LSA_NTT_HEAD_ROUNDS != 2
#define LSA_BF_U " LSA_NTT_HEAD_ROUNDS=" LSA_BF_STR(LSA_NTT_HEAD_ROUNDS)
#else
#define LSA_BF_U ""
#endif
#ifdef LSA_R16_BARRIERS
#define LSA_BF_X " LSA_R16_BARRIERS"
#else
#define LSA_BF_X ""
#endif
#if defined(LSA_R16_WAVES) && LSA_R16_WAVES != 4
#define LSA_BF_Y " LSA_R16_WAVES=" LSA_BF_STR(LSA_R16_WAVES)
#else
#define LSA_BF_Y ""
#endif
#ifdef LSA_NTT_NO_LAZY
#define LSA_BF_Z " LSA_NTT_NO_LAZY"
#else
#define LSA_BF_Z ""
#endif
#ifdef LSA_SHOUP_SINGLE_ASM
#define LSA_BF_AA " LSA_SHOUP_SINGLE_ASM"
#else
#define LSA_BF_AA ""
#endif
#ifdef LSA_AB_DPP_LO
#define LSA_BF_AB " LSA_AB_DPP_LO"
#else
#define LSA_BF_AB ""
#endif
#if defined(LSA_KSMAC_WAVES) && LSA_KSMAC_WAVES != 2
#define LSA_BF_AC " LSA_KSMAC_WAVES=" LSA_BF_STR(LSA_KSMAC_WAVES)
#else
#define LSA_BF_AC ""
#endif
#if defined(LSA_KSMAC_WAVES_FP) && LSA_KSMAC_WAVES_FP != 2
#define LSA_BF_AE " LSA_KSMAC_WAVES_FP=" LSA_BF_STR(LSA_KSMAC_WAVES_FP)
#else
#define LSA_BF_AE ""
#endif
#if defined(LSA_KSMAC_CHUNK) && LSA_KSMAC_CHUNK != 4
#define LSA_BF_AF " LSA_KSMAC_CHUNK=" LSA_BF_STR(LSA_KSMAC_CHUNK)
#else
#define LSA_BF_AF ""
#endif
#ifdef LSA_NTT_DIAG_TW8
#define LSA_BF_AD " LSA_NTT_DIAG_TW8"
#else
#define LSA_BF_AD ""
#endif
#ifdef LSA_VARIANT_NAME
#define LSA_BF_V " variant=" LSA_BF_STR(LSA_VARIANT_NAME)
#else
#define LSA_BF_V ""
#endif
#ifdef LSA_AB_SWITCH   // free-form switch for one-off A/B experiments (named on the command line, e.g. -DLSA_AB_SWITCH=dpp_tail)
#define LSA_BF_W " LSA_AB_SWITCH=" LSA_BF_STR(LSA_AB_SWITCH)
#else
#define LSA_BF_W ""
#endif

// (leading blank stripped by lsa_build_flags)
#define LSA_BUILD_FLAGS_TEXT                                                                                              \
    LSA_BF_DIAG LSA_BF_A LSA_BF_B LSA_BF_C LSA_BF_D LSA_BF_E LSA_BF_F LSA_BF_G LSA_BF_H LSA_BF_I LSA_BF_J LSA_BF_K LSA_BF_L \
        LSA_BF_M LSA_BF_N LSA_BF_O LSA_BF_P LSA_BF_Q LSA_BF_R LSA_BF_S LSA_BF_T LSA_BF_U LSA_BF_V LSA_BF_W LSA_BF_X LSA_BF_Y LSA_BF_Z LSA_BF_AA LSA_BF_AB LSA_BF_AC LSA_BF_AD LSA_BF_AE LSA_BF_AF
