// gfx950 (MI355X, CDNA4) kernels of the APAP moving-DLT engine and their launchers
// (the "_device" half of include/apap_hip.h).
//
// Data layout in HBM
//   table     [n][32]  f64   per keypoint: 30 moment products + (src_x, src_y)
//   vertices  [cells][2] f64 cell sample points
//   moments   [splits][30][cells_pad] f64  A^T W^2 A of every cell, one slab per
//                                          keypoint split; SoA so that lanes = cells
//                                          read and write 512 contiguous bytes
//   H         [cells][9] f32
//   Hinv      [cells][10] f64 = float32 inverses widened once (80 B: five 16-B loads, no per-pixel cvt)
//   lut       [final_h + final_w] i32  canvas row -> cell row, canvas column -> cell column
//
// Compiled with -ffp-contract=off: every fused multiply-add below is written fma().
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <vector>

#include "apap_internal.h"

namespace {

using apap::kMoments;
constexpr int kWave = 64;

// Batched solve: blockIdx.z selects the image pair; element strides between the pairs'
// arrays (0 = shared by all pairs, e.g. one mesh for a batch of equally sized pairs).
struct BatchStride {
    long long table, vertices, moments, denorm, H;
};

// weight of one keypoint for one cell: max(exp(-|v - s| / sigma^2), gamma), float64 like
// apap.py:150-152 (np.sqrt and np.exp on float64).
//
// sqrt and exp are the device library's algorithms (v_rsq_f64 + Goldschmidt with two
// residual corrections; Cody-Waite reduction + degree-11 polynomial + v_ldexp_f64, same
// constants and operation order, hence the same bits) with their range guards replaced
// by what this call site needs: 10 + 18 instructions instead of 17 + 22.
//   * d2 is clamped below at 1e-300 instead of special-casing 0: sqrt gives 1e-150 and
//     exp(-1e-150/sigma^2) == 1.0 exactly, the value for distance 0.
//   * the exponent argument is clamped at -1100 (exp underflows to 0 there).
//   * inf/NaN coordinates are not propagated as NaN (the reference would produce NaN
//     matrices for them).
__device__ __forceinline__ double sqrt_pos(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    return fma(d, h, g);
}

__device__ __forceinline__ double exp_nonpos(double x) {
    x = fmax(x, -1100.0);
    const double n = __builtin_rint(x * 0x1.71547652b82fep+0);       // log2(e)
    double r = fma(n, -0x1.62e42fefa39efp-1, x);                        // -ln2 high
    r = fma(n, -0x1.abc9e3b39803fp-56, r);                              // -ln2 low
    double p = fma(r, 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
    p = fma(r, p, 0x1.71dee623fde64p-19);
    p = fma(r, p, 0x1.a01997c89e6b0p-16);
    p = fma(r, p, 0x1.a01a014761f6ep-13);
    p = fma(r, p, 0x1.6c16c1852b7b0p-10);
    p = fma(r, p, 0x1.1111111122322p-7);
    p = fma(r, p, 0x1.55555555502a1p-5);
    p = fma(r, p, 0x1.5555555555511p-3);
    p = fma(r, p, 0x1.000000000000bp-1);
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    return __builtin_ldexp(p, (int)n);
}

__device__ __forceinline__ double cell_weight(double vx, double vy, double sx, double sy,
                                              double inv_sigma, double gamma) {
    const double dx = vx - sx;
    const double dy = vy - sy;
    const double d2 = fmax(dx * dx + dy * dy, 1e-300);
    const double dist = sqrt_pos(d2);
    const double w = exp_nonpos(-(dist * inv_sigma));
    return fmax(w, gamma);
}

// w^2 for K1, where the weight itself is not an output: 31 fp64 instructions instead of
// 37.  max(exp(-t), gamma)^2 = max(exp(-2t), gamma^2); the squared distance is one
// multiply-add chain that carries the 1e-300 guard; sqrt keeps one residual correction
// (error < 1 ulp instead of correctly rounded).  Each change moves w^2 by at most a few
// ulp of float64 (1e-16 relative) - eight orders below what the float32 H can see; the
// weight tensor that callers can ask for still comes from cell_weight().
//   inv_sigma2 = 2 / sigma^2,  gamma2 = gamma > 0 ? gamma * gamma : 0
__device__ __forceinline__ double cell_weight_sq(double vx, double vy, double sx, double sy,
                                                 double inv_sigma2, double gamma2) {
    const double dx = vx - sx;
    const double dy = vy - sy;
    const double x = fma(dx, dx, fma(dy, dy, 1e-300));
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    const double h = y * 0.5;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    const double d = fma(-g, g, x);
    g = fma(d, h, g);  // h keeps the seed's ~2^-23 error: invisible in a ~2^-45 g correction
    return fmax(exp_nonpos(-(g * inv_sigma2)), gamma2);
}

// Table-driven exp for the MFMA variant, argument pre-scaled: the caller passes
// yu = u * 512 log2(e) >= 0 and gets exp(-u) = 2^(-yu / 512).  With -floor(yu) = 512 m + j
// (one saturating v_cvt_i32_f64 of -yu) and f = fract(yu) in [0, 1):
//     exp(-u) = 2^m * 2^(j / 512) * exp(-f ln2 / 512)
// with the 512 correctly rounded values of 2^(j / 512) in LDS (4 KiB) and a degree-4 polynomial
// in f (truncation (ln2/512)^5 / 120 < 2^-54).  9 fp64 issue slots + 3 integer instructions + one
// ds_read_b64, against 18 slots for the polynomial form of exp_nonpos.  No clamp is needed: a
// huge yu saturates the conversion (2^m underflows to 0) and yu = inf turns into a NaN that the
// caller's fmax(., gamma^2) drops.  Error < 1.5 ulp (1 + u): the u term is the rounding of yu,
// the same size as the rounding of u itself in any evaluation of exp(-u).
// A/B at C3 on one box, K1 duration: polynomial exp 202 us; 64-entry table, degree 5, rndne/cvt
// range reduction 194 us; 256 entries, degree 4, magic-number reduction 180 us; this form 175 us.
constexpr int kExpBits = 9;
constexpr int kExpN = 1 << kExpBits;
constexpr double kExpScale = (double)kExpN * 0x1.71547652b82fep+0;  // 512 log2(e)
__constant__ double kExp2Tab[kExpN] = {
    0x1.0000000000000p+0, 0x1.0058c86da1c0ap+0, 0x1.00b1afa5abcbfp+0, 0x1.010ab5b2cbd11p+0,
    0x1.0163da9fb3335p+0, 0x1.01bd1e77170b4p+0, 0x1.02168143b0281p+0, 0x1.027003103b10ep+0,
    0x1.02c9a3e778061p+0, 0x1.032363d42b027p+0, 0x1.037d42e11bbccp+0, 0x1.03d7411915a8ap+0,
    0x1.04315e86e7f85p+0, 0x1.048b9b35659d8p+0, 0x1.04e5f72f654b1p+0, 0x1.0540727fc1762p+0,
    0x1.059b0d3158574p+0, 0x1.05f5c74f0bec2p+0, 0x1.0650a0e3c1f89p+0, 0x1.06ab99fa6407cp+0,
    0x1.0706b29ddf6dep+0, 0x1.0761ead925493p+0, 0x1.07bd42b72a836p+0, 0x1.0818ba42e7d30p+0,
    0x1.0874518759bc8p+0, 0x1.08d0088f8093fp+0, 0x1.092bdf66607e0p+0, 0x1.0987d61701716p+0,
    0x1.09e3ecac6f383p+0, 0x1.0a402331b9715p+0, 0x1.0a9c79b1f3919p+0, 0x1.0af8f03834e52p+0,
    0x1.0b5586cf9890fp+0, 0x1.0bb23d833d93fp+0, 0x1.0c0f145e46c85p+0, 0x1.0c6c0b6bdae53p+0,
    0x1.0cc922b7247f7p+0, 0x1.0d265a4b520bap+0, 0x1.0d83b23395decp+0, 0x1.0de12a7b26300p+0,
    0x1.0e3ec32d3d1a2p+0, 0x1.0e9c7c55189c6p+0, 0x1.0efa55fdfa9c5p+0, 0x1.0f58503328e6dp+0,
    0x1.0fb66affed31bp+0, 0x1.1014a66f951cep+0, 0x1.1073028d7233ep+0, 0x1.10d17f64d9ef1p+0,
    0x1.11301d0125b51p+0, 0x1.118edb6db2dc1p+0, 0x1.11edbab5e2ab6p+0, 0x1.124cbae51a5c8p+0,
    0x1.12abdc06c31ccp+0, 0x1.130b1e264a0e9p+0, 0x1.136a814f204abp+0, 0x1.13ca058cbae1ep+0,
    0x1.1429aaea92de0p+0, 0x1.1489717425438p+0, 0x1.14e95934f312ep+0, 0x1.154962388149ep+0,
    0x1.15a98c8a58e51p+0, 0x1.1609d83606e12p+0, 0x1.166a45471c3c2p+0, 0x1.16cad3c92df73p+0,
    0x1.172b83c7d517bp+0, 0x1.178c554eaea89p+0, 0x1.17ed48695bbc0p+0, 0x1.184e5d23816c9p+0,
    0x1.18af9388c8deap+0, 0x1.1910eba4df41fp+0, 0x1.1972658375d2fp+0, 0x1.19d4013041dc2p+0,
    0x1.1a35beb6fcb75p+0, 0x1.1a979e2363cf8p+0, 0x1.1af99f8138a1cp+0, 0x1.1b5bc2dc40bf0p+0,
    0x1.1bbe084045cd4p+0, 0x1.1c206fb91588fp+0, 0x1.1c82f95281c6bp+0, 0x1.1ce5a51860746p+0,
    0x1.1d4873168b9aap+0, 0x1.1dab6358e15e8p+0, 0x1.1e0e75eb44027p+0, 0x1.1e71aad999e82p+0,
    0x1.1ed5022fcd91dp+0, 0x1.1f387bf9cda38p+0, 0x1.1f9c18438ce4dp+0, 0x1.1fffd7190241ep+0,
    0x1.2063b88628cd6p+0, 0x1.20c7bc96ffc18p+0, 0x1.212be3578a819p+0, 0x1.21902cd3d09b9p+0,
    0x1.21f49917ddc96p+0, 0x1.2259282fc1f27p+0, 0x1.22bdda27912d1p+0, 0x1.2322af0b63bffp+0,
    0x1.2387a6e756238p+0, 0x1.23ecc1c78903ap+0, 0x1.2451ffb82140ap+0, 0x1.24b760c547f15p+0,
    0x1.251ce4fb2a63fp+0, 0x1.25828c65fa1ffp+0, 0x1.25e85711ece75p+0, 0x1.264e450b3cb82p+0,
    0x1.26b4565e27cddp+0, 0x1.271a8b16f0a30p+0, 0x1.2780e341ddf29p+0, 0x1.27e75eeb3ab98p+0,
    0x1.284dfe1f56381p+0, 0x1.28b4c0ea83f36p+0, 0x1.291ba7591bb70p+0, 0x1.2982b17779965p+0,
    0x1.29e9df51fdee1p+0, 0x1.2a5130f50d65cp+0, 0x1.2ab8a66d10f13p+0, 0x1.2b203fc675d1fp+0,
    0x1.2b87fd0dad990p+0, 0x1.2befde4f2e280p+0, 0x1.2c57e39771b2fp+0, 0x1.2cc00cf2f6c18p+0,
    0x1.2d285a6e4030bp+0, 0x1.2d90cc15d5346p+0, 0x1.2df961f641589p+0, 0x1.2e621c1c14833p+0,
    0x1.2ecafa93e2f56p+0, 0x1.2f33fd6a454d2p+0, 0x1.2f9d24abd886bp+0, 0x1.300670653dfe4p+0,
    0x1.306fe0a31b715p+0, 0x1.30d975721b004p+0, 0x1.31432edeeb2fdp+0, 0x1.31ad0cf63eeacp+0,
    0x1.32170fc4cd831p+0, 0x1.3281375752b40p+0, 0x1.32eb83ba8ea32p+0, 0x1.3355f4fb45e20p+0,
    0x1.33c08b26416ffp+0, 0x1.342b46484ebb4p+0, 0x1.3496266e3fa2dp+0, 0x1.35012ba4ea77dp+0,
    0x1.356c55f929ff1p+0, 0x1.35d7a577dd72bp+0, 0x1.36431a2de883bp+0, 0x1.36aeb428335b4p+0,
    0x1.371a7373aa9cbp+0, 0x1.3786581d3f669p+0, 0x1.37f26231e754ap+0, 0x1.385e91be9c811p+0,
    0x1.38cae6d05d866p+0, 0x1.393761742d808p+0, 0x1.39a401b7140efp+0, 0x1.3a10c7a61d55bp+0,
    0x1.3a7db34e59ff7p+0, 0x1.3aeac4bcdf3eap+0, 0x1.3b57fbfec6cf4p+0, 0x1.3bc559212ef89p+0,
    0x1.3c32dc313a8e5p+0, 0x1.3ca0853c10f28p+0, 0x1.3d0e544ede173p+0, 0x1.3d7c4976d27fap+0,
    0x1.3dea64c123422p+0, 0x1.3e58a63b0a09bp+0, 0x1.3ec70df1c5175p+0, 0x1.3f359bf29743fp+0,
    0x1.3fa4504ac801cp+0, 0x1.40132b07a35dfp+0, 0x1.40822c367a024p+0, 0x1.40f153e4a136ap+0,
    0x1.4160a21f72e2ap+0, 0x1.41d016f44d8f5p+0, 0x1.423fb2709468ap+0, 0x1.42af74a1af3f1p+0,
    0x1.431f5d950a897p+0, 0x1.438f6d5817663p+0, 0x1.43ffa3f84b9d4p+0, 0x1.4470018321a1ap+0,
    0x1.44e086061892dp+0, 0x1.4551318eb43ecp+0, 0x1.45c2042a7d232p+0, 0x1.4632fde7006f4p+0,
    0x1.46a41ed1d0057p+0, 0x1.471566f8827d0p+0, 0x1.4786d668b3237p+0, 0x1.47f86d3001fe5p+0,
    0x1.486a2b5c13cd0p+0, 0x1.48dc10fa920a1p+0, 0x1.494e1e192aed2p+0, 0x1.49c052c5916c4p+0,
    0x1.4a32af0d7d3dep+0, 0x1.4aa532feaada6p+0, 0x1.4b17dea6db7d7p+0, 0x1.4b8ab213d5283p+0,
    0x1.4bfdad5362a27p+0, 0x1.4c70d073537cap+0, 0x1.4ce41b817c114p+0, 0x1.4d578e8bb586bp+0,
    0x1.4dcb299fddd0dp+0, 0x1.4e3eeccbd7b2ap+0, 0x1.4eb2d81d8abffp+0, 0x1.4f26eba2e35f0p+0,
    0x1.4f9b2769d2ca7p+0, 0x1.500f8b804f127p+0, 0x1.508417f4531eep+0, 0x1.50f8ccd3deb0dp+0,
    0x1.516daa2cf6642p+0, 0x1.51e2b00da3b14p+0, 0x1.5257de83f4eefp+0, 0x1.52cd359dfd53dp+0,
    0x1.5342b569d4f82p+0, 0x1.53b85df598d78p+0, 0x1.542e2f4f6ad27p+0, 0x1.54a4298571b06p+0,
    0x1.551a4ca5d920fp+0, 0x1.559098bed1bdfp+0, 0x1.56070dde910d2p+0, 0x1.567dac1351819p+0,
    0x1.56f4736b527dap+0, 0x1.576b63f4d854cp+0, 0x1.57e27dbe2c4cfp+0, 0x1.5859c0d59ca07p+0,
    0x1.58d12d497c7fdp+0, 0x1.5948c32824135p+0, 0x1.59c0827ff07ccp+0, 0x1.5a386b5f43d92p+0,
    0x1.5ab07dd485429p+0, 0x1.5b28b9ee20d1ep+0, 0x1.5ba11fba87a03p+0, 0x1.5c19af482fc8fp+0,
    0x1.5c9268a5946b7p+0, 0x1.5d0b4be135accp+0, 0x1.5d84590998b93p+0, 0x1.5dfd902d47c65p+0,
    0x1.5e76f15ad2148p+0, 0x1.5ef07ca0cbf0fp+0, 0x1.5f6a320dceb71p+0, 0x1.5fe411b078d26p+0,
    0x1.605e1b976dc09p+0, 0x1.60d84fd15612ap+0, 0x1.6152ae6cdf6f4p+0, 0x1.61cd3778bc944p+0,
    0x1.6247eb03a5585p+0, 0x1.62c2c91c56acdp+0, 0x1.633dd1d1929fdp+0, 0x1.63b90532205d8p+0,
    0x1.6434634ccc320p+0, 0x1.64afec30678b7p+0, 0x1.652b9febc8fb7p+0, 0x1.65a77e8dcc390p+0,
    0x1.6623882552225p+0, 0x1.669fbcc140be7p+0, 0x1.671c1c70833f6p+0, 0x1.6798a7420a036p+0,
    0x1.68155d44ca973p+0, 0x1.68923e87bfb7ap+0, 0x1.690f4b19e9538p+0, 0x1.698c830a4c8d4p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6a877541ee718p+0, 0x1.6b052fa75173ep+0, 0x1.6b8315a736c75p+0,
    0x1.6c012750bdabfp+0, 0x1.6c7f64b30aa09p+0, 0x1.6cfdcddd47645p+0, 0x1.6d7c62dea2f8ap+0,
    0x1.6dfb23c651a2fp+0, 0x1.6e7a10a38cee8p+0, 0x1.6ef9298593ae5p+0, 0x1.6f786e7ba9fefp+0,
    0x1.6ff7df9519484p+0, 0x1.70777ce1303f6p+0, 0x1.70f7466f42e87p+0, 0x1.71773c4eaa988p+0,
    0x1.71f75e8ec5f74p+0, 0x1.7277ad3ef9011p+0, 0x1.72f8286ead08ap+0, 0x1.7378d02d50b8fp+0,
    0x1.73f9a48a58174p+0, 0x1.747aa5953c849p+0, 0x1.74fbd35d7cbfdp+0, 0x1.757d2df29ce7cp+0,
    0x1.75feb564267c9p+0, 0x1.768069c1a861dp+0, 0x1.77024b1ab6e09p+0, 0x1.7784597eeba8fp+0,
    0x1.780694fde5d3fp+0, 0x1.7888fda749e5dp+0, 0x1.790b938ac1cf6p+0, 0x1.798e56b7fcf03p+0,
    0x1.7a11473eb0187p+0, 0x1.7a94652e958aap+0, 0x1.7b17b0976cfdbp+0, 0x1.7b9b2988fb9ecp+0,
    0x1.7c1ed0130c132p+0, 0x1.7ca2a4456e7a3p+0, 0x1.7d26a62ff86f0p+0, 0x1.7daad5e2850acp+0,
    0x1.7e2f336cf4e62p+0, 0x1.7eb3bedf2e1b9p+0, 0x1.7f3878491c491p+0, 0x1.7fbd5fbab091fp+0,
    0x1.80427543e1a12p+0, 0x1.80c7b8f4abaa9p+0, 0x1.814d2add106d9p+0, 0x1.81d2cb0d1736ap+0,
    0x1.82589994cce13p+0, 0x1.82de968443d9ap+0, 0x1.8364c1eb941f7p+0, 0x1.83eb1bdadb46dp+0,
    0x1.8471a4623c7adp+0, 0x1.84f85b91e07f1p+0, 0x1.857f4179f5b21p+0, 0x1.8606562ab00ecp+0,
    0x1.868d99b4492edp+0, 0x1.87150c27004c2p+0, 0x1.879cad931a436p+0, 0x1.88247e08e1957p+0,
    0x1.88ac7d98a6699p+0, 0x1.8934ac52be8f7p+0, 0x1.89bd0a478580fp+0, 0x1.8a4597875c644p+0,
    0x1.8ace5422aa0dbp+0, 0x1.8b574029db01ep+0, 0x1.8be05bad61778p+0, 0x1.8c69a6bdb5598p+0,
    0x1.8cf3216b5448cp+0, 0x1.8d7ccbc6c19e6p+0, 0x1.8e06a5e0866d9p+0, 0x1.8e90afc931857p+0,
    0x1.8f1ae99157736p+0, 0x1.8fa553499284bp+0, 0x1.902fed0282c8ap+0, 0x1.90bab6ccce12cp+0,
    0x1.9145b0b91ffc6p+0, 0x1.91d0dad829e70p+0, 0x1.925c353aa2fe2p+0, 0x1.92e7bff148396p+0,
    0x1.93737b0cdc5e5p+0, 0x1.93ff669e2802bp+0, 0x1.948b82b5f98e5p+0, 0x1.9517cf65253d1p+0,
    0x1.95a44cbc8520fp+0, 0x1.9630faccf9243p+0, 0x1.96bdd9a7670b3p+0, 0x1.974ae95cba768p+0,
    0x1.97d829fde4e50p+0, 0x1.98659b9bddb5bp+0, 0x1.98f33e47a22a2p+0, 0x1.9981121235681p+0,
    0x1.9a0f170ca07bap+0, 0x1.9a9d4d47f2598p+0, 0x1.9b2bb4d53fe0dp+0, 0x1.9bba4dc5a3dd3p+0,
    0x1.9c49182a3f090p+0, 0x1.9cd81414380f2p+0, 0x1.9d674194bb8d5p+0, 0x1.9df6a0bcfc15ep+0,
    0x1.9e86319e32323p+0, 0x1.9f15f4499c647p+0, 0x1.9fa5e8d07f29ep+0, 0x1.a0360f4424fcbp+0,
    0x1.a0c667b5de565p+0, 0x1.a156f23701b15p+0, 0x1.a1e7aed8eb8bbp+0, 0x1.a2789dacfe68cp+0,
    0x1.a309bec4a2d33p+0, 0x1.a39b1231475f7p+0, 0x1.a42c980460ad8p+0, 0x1.a4be504f696b1p+0,
    0x1.a5503b23e255dp+0, 0x1.a5e25893523d4p+0, 0x1.a674a8af46052p+0, 0x1.a7072b8950a73p+0,
    0x1.a799e1330b358p+0, 0x1.a82cc9be14dcap+0, 0x1.a8bfe53c12e59p+0, 0x1.a95333beb0b7ep+0,
    0x1.a9e6b5579fdbfp+0, 0x1.aa7a6a1897fd2p+0, 0x1.ab0e521356ebap+0, 0x1.aba26d59a09eep+0,
    0x1.ac36bbfd3f37ap+0, 0x1.accb3e100301ep+0, 0x1.ad5ff3a3c2774p+0, 0x1.adf4dcca5a413p+0,
    0x1.ae89f995ad3adp+0, 0x1.af1f4a17a4735p+0, 0x1.afb4ce622f2ffp+0, 0x1.b04a868742ee4p+0,
    0x1.b0e07298db666p+0, 0x1.b17692a8fa8cdp+0, 0x1.b20ce6c9a8952p+0, 0x1.b2a36f0cf3f3ap+0,
    0x1.b33a2b84f15fbp+0, 0x1.b3d11c43bbd62p+0, 0x1.b468415b749b1p+0, 0x1.b4ff9ade433c6p+0,
    0x1.b59728de5593ap+0, 0x1.b62eeb6ddfc87p+0, 0x1.b6c6e29f1c52ap+0, 0x1.b75f0e844bfc6p+0,
    0x1.b7f76f2fb5e47p+0, 0x1.b89004b3a7804p+0, 0x1.b928cf22749e4p+0, 0x1.b9c1ce8e77680p+0,
    0x1.ba5b030a1064ap+0, 0x1.baf46ca7a67a7p+0, 0x1.bb8e0b79a6f1fp+0, 0x1.bc27df9285775p+0,
    0x1.bcc1e904bc1d2p+0, 0x1.bd5c27e2cb5e5p+0, 0x1.bdf69c3f3a207p+0, 0x1.be91462c95b60p+0,
    0x1.bf2c25bd71e09p+0, 0x1.bfc73b0468d30p+0, 0x1.c06286141b33dp+0, 0x1.c0fe06ff301f4p+0,
    0x1.c199bdd85529cp+0, 0x1.c235aab23e61ep+0, 0x1.c2d1cd9fa652cp+0, 0x1.c36e26b34e065p+0,
    0x1.c40ab5fffd07ap+0, 0x1.c4a77b9881650p+0, 0x1.c544778fafb22p+0, 0x1.c5e1a9f8630adp+0,
    0x1.c67f12e57d14bp+0, 0x1.c71cb269e601fp+0, 0x1.c7ba88988c933p+0, 0x1.c8589584661a1p+0,
    0x1.c8f6d9406e7b5p+0, 0x1.c99553dfa8313p+0, 0x1.ca3405751c4dbp+0, 0x1.cad2ee13da7cbp+0,
    0x1.cb720dcef9069p+0, 0x1.cc1164b994d23p+0, 0x1.ccb0f2e6d1675p+0, 0x1.cd50b869d8f0fp+0,
    0x1.cdf0b555dc3fap+0, 0x1.ce90e9be12cb9p+0, 0x1.cf3155b5bab74p+0, 0x1.cfd1f95018d17p+0,
    0x1.d072d4a07897cp+0, 0x1.d113e7ba2c38cp+0, 0x1.d1b532b08c968p+0, 0x1.d256b596f948cp+0,
    0x1.d2f87080d89f2p+0, 0x1.d39a638197a3cp+0, 0x1.d43c8eacaa1d6p+0, 0x1.d4def2158a91fp+0,
    0x1.d5818dcfba487p+0, 0x1.d62461eec14bep+0, 0x1.d6c76e862e6d3p+0, 0x1.d76ab3a99745bp+0,
    0x1.d80e316c98398p+0, 0x1.d8b1e7e2d479dp+0, 0x1.d955d71ff6075p+0, 0x1.d9f9ff37adb4ap+0,
    0x1.da9e603db3285p+0, 0x1.db42fa45c4dfdp+0, 0x1.dbe7cd63a8315p+0, 0x1.dc8cd9ab294e4p+0,
    0x1.dd321f301b460p+0, 0x1.ddd79e065807dp+0, 0x1.de7d5641c0658p+0, 0x1.df2347f63c159p+0,
    0x1.dfc97337b9b5fp+0, 0x1.e06fd81a2ece1p+0, 0x1.e11676b197d17p+0, 0x1.e1bd4f11f8220p+0,
    0x1.e264614f5a129p+0, 0x1.e30bad7dcee90p+0, 0x1.e3b333b16ee12p+0, 0x1.e45af3fe592e8p+0,
    0x1.e502ee78b3ff6p+0, 0x1.e5ab2334ac7eep+0, 0x1.e653924676d76p+0, 0x1.e6fc3bc24e350p+0,
    0x1.e7a51fbc74c83p+0, 0x1.e84e3e4933c7ep+0, 0x1.e8f7977cdb740p+0, 0x1.e9a12b6bc3181p+0,
    0x1.ea4afa2a490dap+0, 0x1.eaf503ccd2be5p+0, 0x1.eb9f4867cca6ep+0, 0x1.ec49c80faa594p+0,
    0x1.ecf482d8e67f1p+0, 0x1.ed9f78d802dc2p+0, 0x1.ee4aaa2188510p+0, 0x1.eef616ca06dd6p+0,
    0x1.efa1bee615a27p+0, 0x1.f04da28a52e59p+0, 0x1.f0f9c1cb6412ap+0, 0x1.f1a61cbdf5be7p+0,
    0x1.f252b376bba97p+0, 0x1.f2ff860a70c22p+0, 0x1.f3ac948dd7274p+0, 0x1.f459df15b82acp+0,
    0x1.f50765b6e4540p+0, 0x1.f5b5288633625p+0, 0x1.f6632798844f8p+0, 0x1.f7116302bd526p+0,
    0x1.f7bfdad9cbe14p+0, 0x1.f86e8f32a4b45p+0, 0x1.f91d802243c89p+0, 0x1.f9ccadbdac61dp+0,
    0x1.fa7c1819e90d8p+0, 0x1.fb2bbf4c0ba54p+0, 0x1.fbdba3692d514p+0, 0x1.fc8bc4866e8adp+0,
    0x1.fd3c22b8f71f1p+0, 0x1.fdecbe15f6314p+0, 0x1.fe9d96b2a23d9p+0, 0x1.ff4eaca4391b6p+0};

__device__ __forceinline__ double exp_neg_scaled(double yu, const double *tab /* LDS */) {
    constexpr double L = 0x1.62e42fefa39efp-1 / kExpN;  // ln2 / 512
    const int nk = (int)(-yu);                          // truncates toward zero and saturates
    const double f = __builtin_amdgcn_fract(yu);
    const double t = tab[nk & (kExpN - 1)];
    double p = fma(f, L * L * L * L / 24, -(L * L * L / 6));
    p = fma(f, p, L * L / 2);
    p = fma(f, p, -L);
    p = fma(f, p, 1.0);
    return __builtin_ldexp(t * p, nk >> kExpBits);
}

// w^2 for the MFMA variant; `scaled_inv_sigma2` = kExpScale * 2 / sigma^2.  The square root is
// the rsq seed (~2^-23) times a cubic correction: with g = x y and e = 1 - g y,
// sqrt(x) = g (1 - e)^(-1/2) = g + g e (1/2 + 3/8 e) + O(e^3) -- 5 slots, error < 1 ulp.
__device__ __forceinline__ double cell_weight_sq_tab(double vx, double vy, double sx, double sy,
                                                     double scaled_inv_sigma2, double gamma2, const double *tab) {
    const double dx = vx - sx;
    const double dy = vy - sy;
    const double x = fma(dx, dx, fma(dy, dy, 1e-300));
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    const double e = fma(-g, y, 1.0);
    const double c = fma(e, 0.375, 0.5);
    g = fma(g * e, c, g);
    return fmax(exp_neg_scaled(g * scaled_inv_sigma2, tab), gamma2);
}

// w^2 in float32 for the opt-in tolerance tier (APAP_OPT_WEIGHTS_F32 with APAP_OPT_MOMENTS = 24): v_sqrt_f32 and
// v_exp_f32 (1 ulp each), 9 vector instructions instead of 24.  w^2 carries ~2e-7 relative error: the float32 grid moves by
// at most one ulp in a few per cent of its entries (tests/studies/moments24_study.py), like the 24-sum table itself.
//   neg_scale = -2 log2(e) / sigma^2;  NaN coordinates give gamma^2 (v_max_f32 drops the NaN), as in the float64 chain
__device__ __forceinline__ double cell_weight_sq_f32(float vx, float vy, float2 s, float neg_scale, float gamma2) {
    const float dx = vx - s.x;
    const float dy = vy - s.y;
    const float d = __builtin_amdgcn_sqrtf(__builtin_fmaf(dx, dx, dy * dy));
    return (double)fmaxf(__builtin_amdgcn_exp2f(d * neg_scale), gamma2);
}

// A 24-sum table (apap_host_build_table24) carries this quiet-NaN bit pattern in column 28 of every row; the kernels refuse
// a table of the other layout (their grids come out NaN) instead of reading one layout as the other.
constexpr long long kTable24Marker = 0x7ff8242424242424ll;
__device__ __forceinline__ bool table_is_24(const double *__restrict__ table) {
    return __double_as_longlong(table[28]) == kTable24Marker;
}

// --------------------------------------------------------------------------------
// K1 (VALU variant): lanes = cells, one wave per 64-cell tile and keypoint split (grid.y).
// Per keypoint the 32 table doubles are wave-uniform: they arrive through the scalar
// cache into SGPRs and feed v_fma_f64 as the scalar operand, so the vector unit only
// executes the weight and 30 FMAs.  Grid-level splits go to separate slabs that K2 adds.
// --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_assemble_valu(const double *__restrict__ table, int n,
                                                       const double *__restrict__ vertices, int cells,
                                                       int cells_pad, double gamma2, double inv_sigma2,
                                                       int pts_per_split, double *__restrict__ moments,
                                                       BatchStride bs) {
    __shared__ double s_exp2[kExpN];
    for (int j = threadIdx.x; j < kExpN; j += 256) s_exp2[j] = kExp2Tab[j];
    __syncthreads();
    table += (long long)blockIdx.z * bs.table;
    vertices += (long long)blockIdx.z * bs.vertices;
    moments += (long long)blockIdx.z * bs.moments;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the block's four waves own four different 64-cell tiles and walk the SAME keypoints, so a
    // table row fetched through the scalar cache by one wave is a hit for the other three
    const int cell = (blockIdx.x * 4 + wave) * kWave + lane;
    if (cell - lane >= cells_pad) return;  // wave-uniform; no barrier below
    const int cc = min(cell, cells - 1);
    const double vx = vertices[2 * cc];
    const double vy = vertices[2 * cc + 1];
    const int p0 = min(n, (int)blockIdx.y * pts_per_split);
    const int p1 = min(n, p0 + pts_per_split);
    const double scaled_inv_sigma2 = inv_sigma2 * kExpScale;

    double acc[kMoments];
#pragma unroll
    for (int j = 0; j < kMoments; ++j) acc[j] = 0.0;

    // The keypoint's (x, y) is fetched one iteration ahead: scalar loads return out of
    // order, so the only usable wait is lgkmcnt(0); with x, y already in SGPRs the weight
    // (~25 VALU instructions) runs while this iteration's 30 products are in flight.
    double sx = 0.0, sy = 0.0;
    if (p0 < p1) {
        sx = table[(size_t)p0 * APAP_TABLE_STRIDE + 30];
        sy = table[(size_t)p0 * APAP_TABLE_STRIDE + 31];
    }
    // Drain the prologue's loads here, or the compiler's wait for them lands inside the
    // loop right after the prefetch is issued and serialises it.  0x0070 = vmcnt(0) lgkmcnt(0).
    __builtin_amdgcn_s_waitcnt(0x0070);
    for (int p = p0; p < p1; ++p) {
        const double *__restrict__ row = table + (size_t)p * APAP_TABLE_STRIDE;
        const double *__restrict__ nxt = table + (size_t)min(p + 1, p1 - 1) * APAP_TABLE_STRIDE;
        const double nsx = nxt[30], nsy = nxt[31];
        double r[kMoments];
#pragma unroll
        for (int j = 0; j < kMoments; ++j) r[j] = row[j];
        __builtin_amdgcn_sched_barrier(0);  // all scalar loads of the iteration are issued above the weight
        const double w2 = cell_weight_sq_tab(vx, vy, sx, sy, scaled_inv_sigma2, gamma2, s_exp2);
#pragma unroll
        for (int j = 0; j < kMoments; ++j) acc[j] = fma(w2, r[j], acc[j]);
        __builtin_amdgcn_sched_barrier(0);  // or the copy is hoisted to the loop top with its own lgkmcnt(0)
        sx = nsx;
        sy = nsy;
    }
    double *dst = moments + (size_t)blockIdx.y * kMoments * cells_pad + cell;
#pragma unroll
    for (int j = 0; j < kMoments; ++j) dst[(size_t)j * cells_pad] = acc[j];
}

// --------------------------------------------------------------------------------
// K1 (MFMA variant): the 30 moment sums of 16 cells over 4 keypoints are one
// v_mfma_f64_16x16x4_f64:  D[cell][moment] += A[cell][point] * B[point][moment] with
// A = w^2 (computed by the lane that owns (cell = lane & 15, point = lane >> 4)) and
// B = the keypoint table staged in LDS.  Two MFMAs cover the 32 table columns (30
// moments + the keypoint's x, y, whose sums are ignored).  The matrix pipe does the
// accumulation and the cross-keypoint reduction; the vector unit only computes weights.
// A block is 4 waves = 64 cells sharing each 64-keypoint LDS chunk (16 KiB, double
// buffered, register-staged so the next chunk's global loads fly during the MFMAs).
// --------------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));
// Keypoints per LDS buffer (16 KiB).  32 / 48 / 64 measured the same (196-200 us at C3): the
// kernel is bound by the fp64 pipe, not by barriers or occupancy (VGPR + AGPR = 102 -> 4 waves
// per SIMD; forcing <= 96 registers for 5-6 waves per SIMD changed nothing either).
constexpr int kChunk = 64;
static_assert(kChunk % 16 == 0 && kChunk % 4 == 0, "chunk must split evenly over 256 threads and 4-keypoint steps");

// Byte offset of table entry (row r, column c) in an LDS chunk: 256-B rows; odd rows swap
// their 128-B halves so that the two keypoint rows a 32-lane ds_read_b64 covers (lanes
// 0-15 -> row r, 16-31 -> row r+1) fall on disjoint halves of the 64 banks.
__device__ __forceinline__ int lds_off(int r, int c) { return r * 256 + ((c ^ ((r & 1) << 4)) << 3); }

//
// kM24 (APAP_OPT_MOMENTS = 24, opt-in): the table holds the 24 sums of the exact products (SURVEY.md section 8a:
// A^T W^2 A = [[S0, 0, Sx], [0, S0, Sy], [Sx, Sy, Sr]], 6 unique entries each) in columns 0..23; per step ONE
// v_mfma_f64_16x16x4_f64 covers columns 0..15 and TWO v_mfma_f64_4x4x4_4b_f64 columns 16..19 and 20..23, all three fed by
// the same w^2 register: the two instruction forms share the lane -> (cell, keypoint) mapping of the A operand
// (k_assemble_mfma4 below has the small form's layout).  Not bit-identical to the reference (its products are rounded to
// float32, apap.py:103-119): one float32 ulp in a few per cent of the grid's entries.  kW32: weights by cell_weight_sq_f32.
template <int kWaves, bool kM24, bool kW32>
__global__ __launch_bounds__(kWaves * 64) void k_assemble_mfma(const double *__restrict__ table, int n,
                                                       const double *__restrict__ vertices, int cells,
                                                       int cells_pad, double gamma2, double inv_sigma2,
                                                       int pts_per_split, double *__restrict__ moments,
                                                       BatchStride bs) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][kChunk * 256];
    table += (long long)blockIdx.z * bs.table;
    vertices += (long long)blockIdx.z * bs.vertices;
    moments += (long long)blockIdx.z * bs.moments;
    __shared__ double s_exp2[kExpN];
    // The exp table goes through registers and is written to LDS AFTER the first chunk's loads have been
    // issued: `s_exp2[j] = kExp2Tab[j]` up here made the block wait out one global-memory latency before
    // it even asked for its keypoints (two latencies in a row; it shows on small meshes).
    static_assert(kExpN % (kWaves * 64) == 0 || (kWaves * 64) % kExpN == 0, "exp table vs block size");
    constexpr int kExpPerThread = (kExpN + kWaves * 64 - 1) / (kWaves * 64);
    double exp_stage[kExpPerThread];
#pragma unroll
    for (int i = 0; i < kExpPerThread; ++i) exp_stage[i] = kExp2Tab[(threadIdx.x + i * kWaves * 64) & (kExpN - 1)];
    const double scaled_inv_sigma2 = inv_sigma2 * kExpScale;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int kgrp = lane >> 4;  // which of the step's 4 keypoints this lane weighs
    const int col = lane & 15;   // cell within the wave's 16 (A operand) / table column (B operand)
    const int cell = blockIdx.x * (16 * kWaves) + wave * 16 + col;
    const int cc = min(cell, cells - 1);
    const double vx = vertices[2 * cc];
    const double vy = vertices[2 * cc + 1];
    const float vxf = (float)vx, vyf = (float)vy;
    const float neg_scale_f = (float)(inv_sigma2 * -0x1.71547652b82fep+0), gamma2f = (float)gamma2;
    const int p_begin = min(n, (int)blockIdx.y * pts_per_split);
    const int p_end = min(n, p_begin + pts_per_split);
    const int nchunks = (p_end - p_begin + kChunk - 1) / kChunk;

    // staging: thread t moves 16-byte pieces t, t+256, t+512, t+768 of a chunk;
    // piece q = (row q >> 4, 16-byte slot q & 15)
    constexpr int kThreads = kWaves * 64;
    constexpr int kPieces = kChunk * 16 / kThreads;  // 16-byte pieces of a chunk per thread
    double2 stage[kPieces];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
            const int q = tid + kThreads * i;
            const int p = p_begin + c * kChunk + (q >> 4);
            stage[i] = (p < p_end) ? *reinterpret_cast<const double2 *>(table + (size_t)p * APAP_TABLE_STRIDE + 2 * (q & 15))
                                   : make_double2(0.0, 0.0);  // zero rows add nothing whatever their weight
        }
    };
    auto store_chunk = [&](int b) {
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
            const int q = tid + kThreads * i;
            const int r = q >> 4;
            const int slot = (q & 15) ^ ((r & 1) << 3);
            *reinterpret_cast<double2 *>(&lds[b][r * 256 + slot * 16]) = stage[i];
        }
    };

    const int off_xy = kW32 ? lds_off(kgrp, 29) : lds_off(kgrp, 30), off_b0 = lds_off(kgrp, col);
    // second operand: columns 16..31 (30 sums), or columns 16 + (lane & 3) and, 32 bytes on, 20 + (lane & 3) (24 sums)
    const int off_b1 = lds_off(kgrp, 16 + (kM24 ? (lane & 3) : col));
    double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    double acc_a = 0.0, acc_b = 0.0;
    auto weight = [&](const unsigned char *at) -> double {
        if constexpr (kW32) {
            return cell_weight_sq_f32(vxf, vyf, *reinterpret_cast<const float2 *>(at), neg_scale_f, gamma2f);
        } else {
            const double2 xy = *reinterpret_cast<const double2 *>(at);
            return cell_weight_sq_tab(vx, vy, xy.x, xy.y, scaled_inv_sigma2, gamma2, s_exp2);
        }
    };
    auto accumulate = [&](double w2, const unsigned char *at0, const unsigned char *at1) {
        const double b0 = *reinterpret_cast<const double *>(at0);
        const double b1 = *reinterpret_cast<const double *>(at1);
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w2, b0, acc0, 0, 0, 0);
        if constexpr (kM24) {
            acc_a = __builtin_amdgcn_mfma_f64_4x4x4f64(w2, b1, acc_a, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f64_4x4x4f64(w2, *reinterpret_cast<const double *>(at1 + 32), acc_b, 0, 0, 0);
        } else {
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w2, b1, acc1, 0, 0, 0);
        }
    };
    if (nchunks > 0) load_chunk(0);
#pragma unroll
    for (int i = 0; i < kExpPerThread; ++i) s_exp2[(tid + i * kThreads) & (kExpN - 1)] = exp_stage[i];  // visible after the barrier
    if (nchunks > 0) store_chunk(0);
    __syncthreads();
    // Drain the prologue's loads (vertex, first chunk) here: otherwise the compiler's wait
    // for the vertex registers lands inside the step loop as vmcnt(0) and also waits for
    // the NEXT chunk's prefetch, which is meant to fly during the MFMAs.
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
    // Whole chunks first, in a loop with ONE exit (round 6: with the partial chunk's early `break` inside it the compiler kept more
    // copies of the accumulators); a partial last chunk after it runs only the steps that hold keypoints (its missing rows are
    // zero and would add +0 to every sum, at the full price of a step: C3's 1000 keypoints per split are 15 chunks and 10 steps
    // of the 16th - 2.3 % of the kernel).  Same MFMAs in the same order as ever: identical sums.
    // (Also round 6, not kept: ONE block of 8 waves running both keypoint splits of its 64 cells and adding them through LDS
    // before it stores one slab - bit-identical, K2 8.4 -> 7.2 us, but K1 147.5 -> 173 us: eight waves on one barrier per
    // chunk, two 68 KB blocks per CU.  profiles/r06_k1_compiler.txt.)
    const int nfull = (p_end - p_begin) / kChunk;
    for (int c = 0; c < nfull; ++c) {
        if (c + 1 < nchunks) load_chunk(c + 1);
        const unsigned char *buf = lds[c & 1];
        // The weights of g steps first, then their 2 g MFMAs back to back: g times fewer MFMA <-> VALU
        // transitions (~10 issue cycles each, profiles/r02_coexec.txt) and g independent weight chains
        // for the scheduler.  Same MFMAs in the same order per accumulator: bit-identical sums.
        // A/B at C3 on one box (tools/ab_build.sh): g = 1: 169.5-170.7 us, 2: 165.5-166.3, 4: 165.5-166.1,
        // 8: 163.8-164.1, 16: 163.9-164.2.  (s_setprio 1 / 3 around the MFMA group: +-1 %; a 128-keypoint
        // chunk: +4 %; profiles/r02_k1_variants.txt.)
#pragma unroll
        for (int s0 = 0; s0 < kChunk / 4; s0 += 8) {
            double w2[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) w2[g] = weight(buf + off_xy + 1024 * (s0 + g));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 8; ++g) accumulate(w2[g], buf + off_b0 + 1024 * (s0 + g), buf + off_b1 + 1024 * (s0 + g));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c + 1 < nchunks) store_chunk((c + 1) & 1);
        __syncthreads();
    }
    if (nfull < nchunks) {      // the partial last chunk (staged and published by the loop's last pass, or by the prologue)
        const unsigned char *buf = lds[nfull & 1];
        const int steps_here = (p_end - p_begin - nfull * kChunk + 3) >> 2;
        for (int s1 = 0; s1 < steps_here; ++s1)
            accumulate(weight(buf + off_xy + 1024 * s1), buf + off_b0 + 1024 * s1, buf + off_b1 + 1024 * s1);
    }

    // D layout of v_mfma_f64_16x16x4_f64: register i of lane l is D[row = (l >> 4) + 4 i][col = l & 15],
    // row = cell within the wave's 16, col = moment index (acc0: 0..15, acc1: 16..31).
    constexpr int kSums = kM24 ? 24 : kMoments;
    double *slab = moments + (size_t)blockIdx.y * kSums * cells_pad;
    const int cell_base = blockIdx.x * (16 * kWaves) + wave * 16 + kgrp;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ci = cell_base + 4 * i;
        if (ci < cells_pad) {  // an 8-wave block may overhang the padded cell count
            // (plain stores: non-temporal ones were measured - K1 150.0 -> 148.7 us, but the 128-byte pieces of a moment row no
            // longer merge in the L2 and WRITE_SIZE grows from 18.8 to 34.1 MB per launch, profiles/r05_solve_experiments.txt)
            slab[(size_t)col * cells_pad + ci] = acc0[i];
            if (!kM24 && col < kMoments - 16) slab[(size_t)(16 + col) * cells_pad + ci] = acc1[i];
        }
    }
    if constexpr (kM24) {
        // D[b][i][j] of v_mfma_f64_4x4x4_4b_f64 sits in lane 16 i + 4 b + j: cell 4 b + i of the wave's 16, column 16 + j / 20 + j
        const int ci = blockIdx.x * (16 * kWaves) + wave * 16 + 4 * ((lane >> 2) & 3) + (lane >> 4);
        if (ci < cells_pad) {
            slab[(size_t)(16 + (lane & 3)) * cells_pad + ci] = acc_a;
            slab[(size_t)(20 + (lane & 3)) * cells_pad + ci] = acc_b;
        }
    }
}

// --------------------------------------------------------------------------------
// K1 (small-MFMA variant): the same lane -> (cell, keypoint) mapping, accumulated by
// v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4: block b = cells 4b..4b+3 of a 16-cell group, j = 4
// table columns).  Operand layout (tools/mfma4_layout.hip): A[b][i][k] in lane 16k + 4b + i,
// B[b][k][j] in lane 16k + 4b + j, D[b][i][j] in lane 16i + 4b + j.  Eight instructions cover the
// 32 table columns; a wave owns kGroups 16-cell groups that share every B operand.
// Why it exists: tools/coexec.hip shows that NO vector instruction co-executes with an f64 MFMA on
// this chip - issue time simply adds up - and that in a pure stream the 16x16x4 form costs ~104
// cycles per instruction where four 4x4x4_4b cost 4 x 16.3 (profiles/r02_coexec.txt).  Measured in
// the kernel (profiles/r02_k1_variants.txt): C3 178-182 us with 16 cells per wave, 163-168 us with
// 32 (one more slab for K2), against 167-170 us for the 16x16x4 kernel and 172-174 us for the VALU
// kernel: every form ends at ~266 issue cycles per 64 (cell, keypoint) pairs = 128 for the
// accumulation (32 x 4) + the weight chain.  The fp64 pipe is the bound, not the MFMA shape.
// --------------------------------------------------------------------------------
// kQuads = 8: the 32 columns of the 30-sum table; kQuads = 6: columns 0..23 of the 24-sum table (APAP_OPT_MOMENTS = 24, see
// k_assemble_mfma); kW32: float32 weights (cell_weight_sq_f32).
template <int kGroups, int kQuads, bool kW32>
__global__ __launch_bounds__(256) void k_assemble_mfma4(const double *__restrict__ table, int n,
                                                        const double *__restrict__ vertices, int cells,
                                                        int cells_pad, double gamma2, double inv_sigma2,
                                                        int pts_per_split, double *__restrict__ moments,
                                                        BatchStride bs) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][kChunk * 256];
    table += (long long)blockIdx.z * bs.table;
    vertices += (long long)blockIdx.z * bs.vertices;
    moments += (long long)blockIdx.z * bs.moments;
    __shared__ double s_exp2[kExpN];
    for (int j = threadIdx.x; j < kExpN; j += 256) s_exp2[j] = kExp2Tab[j];  // visible after the first barrier
    const double scaled_inv_sigma2 = inv_sigma2 * kExpScale;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int kgrp = lane >> 4;  // which of the step's 4 keypoints this lane weighs / reads B for
    const int col = lane & 15;   // cell within a 16-cell group (A operand)
    const int cell0 = (blockIdx.x * 4 + wave) * (16 * kGroups);
    double vx[kGroups], vy[kGroups];
    float vxf[kGroups], vyf[kGroups];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
        const int cc = min(cell0 + 16 * g + col, cells - 1);
        vx[g] = vertices[2 * cc];
        vy[g] = vertices[2 * cc + 1];
        vxf[g] = (float)vx[g];
        vyf[g] = (float)vy[g];
    }
    const float neg_scale_f = (float)(inv_sigma2 * -0x1.71547652b82fep+0), gamma2f = (float)gamma2;
    const int p_begin = min(n, (int)blockIdx.y * pts_per_split);
    const int p_end = min(n, p_begin + pts_per_split);
    const int nchunks = (p_end - p_begin + kChunk - 1) / kChunk;

    constexpr int kPieces = kChunk * 16 / 256;  // 16-byte pieces of a chunk per thread
    double2 stage[kPieces];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
            const int q = tid + 256 * i;
            const int p = p_begin + c * kChunk + (q >> 4);
            stage[i] = (p < p_end) ? *reinterpret_cast<const double2 *>(table + (size_t)p * APAP_TABLE_STRIDE + 2 * (q & 15))
                                   : make_double2(0.0, 0.0);
        }
    };
    auto store_chunk = [&](int b) {
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
            const int q = tid + 256 * i;
            const int r = q >> 4;
            const int slot = (q & 15) ^ ((r & 1) << 3);
            *reinterpret_cast<double2 *>(&lds[b][r * 256 + slot * 16]) = stage[i];
        }
    };

    const int off_xy = kW32 ? lds_off(kgrp, 29) : lds_off(kgrp, 30);
    // column 4 m + j of row kgrp: the swizzle only flips bit 4 of the column, so two per-lane bases
    // (columns 0-15 and 16-31) plus the immediate 32 (m & 3)
    const int off_lo = lds_off(kgrp, lane & 3), off_hi = lds_off(kgrp, 16 + (lane & 3));
    double acc[kGroups][kQuads];
#pragma unroll
    for (int g = 0; g < kGroups; ++g)
#pragma unroll
        for (int m = 0; m < kQuads; ++m) acc[g][m] = 0.0;
    if (nchunks > 0) {
        load_chunk(0);
        store_chunk(0);
    }
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): see k_assemble_mfma
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) load_chunk(c + 1);
        const unsigned char *buf = lds[c & 1];
#pragma unroll
        for (int s = 0; s < kChunk / 4; ++s) {
            double bv[kQuads];
#pragma unroll
            for (int m = 0; m < kQuads; ++m)
                bv[m] = *reinterpret_cast<const double *>(buf + (m < 4 ? off_lo : off_hi) + 32 * (m & 3) + 1024 * s);
            double w2[kGroups];
            if constexpr (kW32) {
                const float2 xyf = *reinterpret_cast<const float2 *>(buf + off_xy + 1024 * s);
#pragma unroll
                for (int g = 0; g < kGroups; ++g) w2[g] = cell_weight_sq_f32(vxf[g], vyf[g], xyf, neg_scale_f, gamma2f);
            } else {
                const double2 xy = *reinterpret_cast<const double2 *>(buf + off_xy + 1024 * s);
#pragma unroll
                for (int g = 0; g < kGroups; ++g)
                    w2[g] = cell_weight_sq_tab(vx[g], vy[g], xy.x, xy.y, scaled_inv_sigma2, gamma2, s_exp2);
            }
#pragma unroll
            for (int m = 0; m < kQuads; ++m)
#pragma unroll
                for (int g = 0; g < kGroups; ++g)
                    acc[g][m] = __builtin_amdgcn_mfma_f64_4x4x4f64(w2[g], bv[m], acc[g][m], 0, 0, 0);
        }
        if (c + 1 < nchunks) store_chunk((c + 1) & 1);
        __syncthreads();
    }
    // D[b][i][j] sits in lane 16 i + 4 b + j: cell 4 b + i of the group, table column 4 m + j
    constexpr int kSums = kQuads == 6 ? 24 : kMoments;
    double *slab = moments + (size_t)blockIdx.y * kSums * cells_pad;
    const int dcell = 4 * ((lane >> 2) & 3) + (lane >> 4);
    const int dj = lane & 3;
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
        const int ci = cell0 + 16 * g + dcell;
        if (ci < cells_pad) {
#pragma unroll
            for (int m = 0; m < kQuads; ++m)
                if (4 * m + dj < kSums) slab[(size_t)(4 * m + dj) * cells_pad + ci] = acc[g][m];
        }
    }
}

// --------------------------------------------------------------------------------
// K2: per-cell symmetric 9x9 eigen-solve, lanes = cells, everything in registers: inverse
// iteration on an L D L^T factorisation by default, cyclic Jacobi sweeps (45 + 81 doubles)
// as the fallback and as the selectable alternative; cells whose normal matrix cannot carry the
// answer (eigen-gap below 1e-3 of the trace, underflowed sums, fewer than 5 keypoints) are re-solved
// from the weighted rows themselves (qr_resolve); then the vector is de-normalised
// (apap.py:161-168) and stored as float32.
// --------------------------------------------------------------------------------
__host__ __device__ constexpr int tri(int i, int j) {
    return i <= j ? i * (19 - i) / 2 + (j - i) : j * (19 - j) / 2 + (i - j);
}

template <int P, int Q>
__device__ __forceinline__ void jacobi_rotate(double (&a)[45], double (&v)[81]) {
    const double apq = a[tri(P, Q)];
    const double app = a[tri(P, P)];
    const double aqq = a[tri(Q, Q)];
    const double theta = 0.5 * (aqq - app) / apq;
    double t = copysign(1.0, theta) / (fabs(theta) + sqrt(fma(theta, theta, 1.0)));
    t = (apq != 0.0) ? t : 0.0;  // also discards the NaN of 0/0
    const double c = 1.0 / sqrt(fma(t, t, 1.0));
    const double s = t * c;
    a[tri(P, P)] = fma(-t, apq, app);
    a[tri(Q, Q)] = fma(t, apq, aqq);
    a[tri(P, Q)] = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (k != P && k != Q) {
            const double akp = a[tri(k, P)];
            const double akq = a[tri(k, Q)];
            a[tri(k, P)] = fma(c, akp, -(s * akq));
            a[tri(k, Q)] = fma(s, akp, c * akq);
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const double vkp = v[9 * k + P];
        const double vkq = v[9 * k + Q];
        v[9 * k + P] = fma(c, vkp, -(s * vkq));
        v[9 * k + Q] = fma(s, vkp, c * vkq);
    }
}

__device__ __forceinline__ void jacobi_sweep(double (&a)[45], double (&v)[81]) {
#define APAP_ROW(p, q) jacobi_rotate<p, q>(a, v);
    APAP_ROW(0, 1) APAP_ROW(0, 2) APAP_ROW(0, 3) APAP_ROW(0, 4) APAP_ROW(0, 5) APAP_ROW(0, 6) APAP_ROW(0, 7) APAP_ROW(0, 8)
    APAP_ROW(1, 2) APAP_ROW(1, 3) APAP_ROW(1, 4) APAP_ROW(1, 5) APAP_ROW(1, 6) APAP_ROW(1, 7) APAP_ROW(1, 8)
    APAP_ROW(2, 3) APAP_ROW(2, 4) APAP_ROW(2, 5) APAP_ROW(2, 6) APAP_ROW(2, 7) APAP_ROW(2, 8)
    APAP_ROW(3, 4) APAP_ROW(3, 5) APAP_ROW(3, 6) APAP_ROW(3, 7) APAP_ROW(3, 8)
    APAP_ROW(4, 5) APAP_ROW(4, 6) APAP_ROW(4, 7) APAP_ROW(4, 8)
    APAP_ROW(5, 6) APAP_ROW(5, 7) APAP_ROW(5, 8)
    APAP_ROW(6, 7) APAP_ROW(6, 8)
    APAP_ROW(7, 8)
#undef APAP_ROW
}

// out = x * y for row-major 3x3, k-ordered FMA chain per element
__device__ __forceinline__ void mul3(const double *x, const double *y, double *out) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            out[3 * i + j] = fma(x[3 * i + 2], y[6 + j], fma(x[3 * i + 1], y[3 + j], x[3 * i] * y[j]));
}

constexpr int kMaxSweeps = 15;

// Eigenvector of the (pick_rank)-th smallest eigenvalue by cyclic Jacobi.  `a` is
// destroyed.  Which eigenvector: the reference takes the LAST row of the V^T that the thin
// SVD of the 2n x 9 system returns, singular values sorted descending (apap.py:160-161).
// For 2n >= 9 that is the smallest eigenvalue of A^T W^2 A (pick_rank 0); for fewer rows
// the thin V^T has only 2n rows and its last one belongs to the (9 - 2n)-th smallest
// eigenvalue.  Rank = number of strictly smaller diagonal entries, ties broken by index.
__device__ __forceinline__ double jacobi_eigvec(double (&a)[45], int pick_rank, double (&h)[9]) {
    double v[81];
#pragma unroll
    for (int k = 0; k < 81; ++k) v[k] = (k % 10 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < kMaxSweeps; ++sweep) {
        jacobi_sweep(a, v);
        double off2 = 0.0, trace = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            trace += fabs(a[tri(i, i)]);
#pragma unroll
            for (int j = i + 1; j < 9; ++j) off2 = fma(a[tri(i, j)], a[tri(i, j)], off2);
        }
        // quadratic convergence: 1e-17 relative is reached one sweep after ~1e-8.
        // NaN input never converges and stops at kMaxSweeps.
        const bool done = off2 <= 1e-34 * trace * trace;
        if (__all(done)) break;
    }
    int best = 0;
    double lam0 = 0.0, lam1 = 0.0;  // the two smallest eigenvalues
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        int rank = 0;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            if (j != i) {
                const double di = a[tri(i, i)], dj = a[tri(j, j)];
                rank += (dj < di || (dj == di && j < i)) ? 1 : 0;
            }
        }
        best = (rank == pick_rank) ? i : best;
        lam0 = (rank == 0) ? a[tri(i, i)] : lam0;
        lam1 = (rank == 1) ? a[tri(i, i)] : lam1;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        double x = v[9 * k];
#pragma unroll
        for (int c = 1; c < 9; ++c) x = (best == c) ? v[9 * k + c] : x;
        h[k] = x;
    }
    return lam1 - lam0;
}

// Smallest eigenvector by inverse iteration on an in-register L D L^T factorisation
// (no pivoting: the matrix is a Gram matrix).  The gap of these systems is huge
// (lambda_9 / lambda_8 ~ 1e-5 on image data, ~1e-3 on the 120-point fixtures), so the
// iteration contracts by that factor per step; it runs until two successive unit vectors
// agree to 1e-14 (sign-adjusted).  Returns false when a pivot is not positive/finite or
// the iteration has not converged in kMaxInvIt steps - the caller then falls back to
// Jacobi, which needs no gap.  ~0.9k instructions against ~30k for six Jacobi sweeps.
constexpr int kMaxInvIt = 12;

__host__ __device__ constexpr int low(int i, int j) { return i * (i - 1) / 2 + j; }  // strict lower, i > j

// 1 / x to <= 1 ulp (v_rcp_f64 + two Newton steps: 5 instructions where the IEEE division sequence is 12).
// inf / NaN / 0 in -> NaN or inf out; the callers test their operands.
__device__ __forceinline__ double rcp_full(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}

// 1 / sqrt(x) to ~1 ulp (v_rsq_f64 + two Newton steps)
__device__ __forceinline__ double rsqrt_full(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = fma(y, fma(-hx * y, y, 0.5), y);
    return fma(y, fma(-hx * y, y, 0.5), y);
}

// The normal matrix is [[S0, 0, S1], [0, S0, S2], [S1^T, S2^T, S3]] (3x3 blocks): rows 3..5 of its unit
// lower factor are zero in columns 0..2 and repeat rows 0..2 next to them, with the same pivots.  The
// factorisation and the two triangular solves below skip the products with those exact zeros (a term
// 0 * t added by an FMA changes nothing) and copy the repeated block: the same numbers as the dense
// recurrences, bit for bit, in 145 instead of ~300 instructions and 27 instead of 36 per solve.
__host__ __device__ constexpr bool lnz(int i, int k) { return !(i >= 3 && i <= 5 && k <= 2); }  // l(i, k) != 0 structurally

__device__ __forceinline__ int count_eigs_below(const double (&m)[kMoments], double mu);

// `below` (when m is given): the number of eigenvalues below (an upper bound of lambda_9) + gap_tol - the
// conditioning guard of eigen_denorm_cell.  It is evaluated at the top of the third solve with the
// Rayleigh bound of the second (exact to ~1e-10 relative by then) so that its ~100 instructions - six
// reciprocals in a row - fill the issue slots that the solve's dependent chains leave empty, instead of
// standing alone after the loop (K2 runs one wave per SIMD: nothing else hides latency there).
__device__ __forceinline__ bool inverse_iteration(const double (&a)[45], double (&h)[9], double &rho,
                                                  const double (&m)[kMoments], double gap_tol, int &below) {
    double l[36], d[9], rd[9];  // unit lower factor, pivots and their reciprocals
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        double t[9];  // t[k] = l(j,k) * d[k]
        if (j >= 3 && j <= 5) {  // the second copy of S0's factor
            d[j] = d[j - 3];
            rd[j] = rd[j - 3];
#pragma unroll
            for (int k = 3; k < 9; ++k)
                if (k < j) t[k] = l[low(j, k)] * d[k];
        } else {
            double dj = a[tri(j, j)];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                if (k < j) {
                    t[k] = l[low(j, k)] * d[k];
                    dj = fma(-l[low(j, k)], t[k], dj);
                }
            }
            ok = ok && (dj > 0.0) && (dj < 1e300);
            d[j] = dj;
            rd[j] = rcp_full(dj);
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            if (i > j) {
                if (!lnz(i, j)) {
                    l[low(i, j)] = 0.0;  // never read
                } else if (i <= 5 && j >= 3) {
                    l[low(i, j)] = l[low(i - 3, j - 3)];
                } else {
                    double sum = a[tri(j, i)];
#pragma unroll
                    for (int k = 0; k < 9; ++k)
                        if (k < j && lnz(i, k) && lnz(j, k)) sum = fma(-l[low(i, k)], t[k], sum);
                    l[low(i, j)] = sum * rd[j];
                }
            }
        }
    }
    double v[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = 1.0 / 3.0;
    bool done = false;
    double vMv = 1.0;  // v^T M^-1 v of the last step: 1 / vMv >= lambda_min (Rayleigh quotient of M^-1)
    below = -1;
    // (fully unrolled by the compiler; a rolled loop - one copy of the body in the instruction cache - was measured:
    // no faster, and the guard at it == 2 then no longer interleaves with the third solve)
    for (int it = 0; it < kMaxInvIt; ++it) {
        if (it == 2) below = count_eigs_below(m, rcp_full(vMv) + gap_tol);
        double y[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {  // L z = v
            double sum = v[i];
#pragma unroll
            for (int k = 0; k < 9; ++k)
                if (k < i && lnz(i, k)) sum = fma(-l[low(i, k)], y[k], sum);
            y[i] = sum;
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) y[i] *= rd[i];  // D
#pragma unroll
        for (int i = 8; i >= 0; --i) {  // L^T y = z
            double sum = y[i];
#pragma unroll
            for (int k = 0; k < 9; ++k)
                if (k > i && lnz(k, i)) sum = fma(-l[low(k, i)], y[k], sum);
            y[i] = sum;
        }
        double nrm2 = 0.0, dot = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            nrm2 = fma(y[i], y[i], nrm2);
            dot = fma(y[i], v[i], dot);
        }
        const double scale = copysign(rsqrt_full(nrm2), dot);
        vMv = fabs(dot);
        // The first two solves cannot have converged (from the uniform start the vector still moves by ~1e-5 in
        // the second): no convergence test there - the loop is fully unrolled, `it` is a constant in each copy.
        double change = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double yn = y[i] * scale;
            if (it >= 2) change = fmax(change, fabs(yn - v[i]));
            v[i] = yn;
        }
        if (it >= 2) {
            done = change <= 1e-14;  // false for NaN; the error left is that times lambda_9 / lambda_8
            if (__all(done || !ok)) break;
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) h[k] = v[k];
    rho = rcp_full(vMv);
    if (below < 0) below = count_eigs_below(m, rho + gap_tol);  // converged in two solves: wave-uniform, practically never
    return ok && done;
}

// ---- conditioning guard -----------------------------------------------------------
// The reference takes the SVD of the weighted 2n x 9 matrix W A itself (apap.py:159-161); K1/K2
// solve A^T W^2 A, whose eigenvector v9 carries an error ~ eps * lambda_1 / (lambda_8 - lambda_9)
// - the SQUARE of what the SVD's eps * sigma_1 / (sigma_8 - sigma_9) is.  On the Hartley-normalised
// systems of image data (lambda_8 - lambda_9) / trace sits at 1.5e-2 (C1-C5, every golden case); with
// gamma = 0 and sigma of a few pixels the weights span 10+ orders of magnitude and it falls to
// 1e-7 ... 1e-20 (soak seeds 544, 659, 795, 814, 883: hundreds of pixels off); nearly determined systems
// (5-6 keypoints) sit at 4e-4 ... 3e-3.  A cell whose gap is below kGapTol * trace is re-solved from the
// weighted rows themselves (qr_resolve below).
constexpr double kGapTol = 1e-3;  // normal-equation error ~ 50 eps / kGapTol = 1e-11 relative at the threshold

__device__ __forceinline__ double rcp_fast(double x) {  // ~2^-46: only signs of pivots are used
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}

// Number of eigenvalues of A^T W^2 A below mu (Sylvester's law of inertia: the negative pivots of
// an L D L^T of M - mu I), using the block structure [[S0, 0, S1], [0, S0, S2], [S1^T, S2^T, S3]]:
// one 3x3 factorisation serves both S0 blocks, then the 3x3 Schur complement.  ~100 instructions.
// A zero or NaN pivot counts as negative (the caller then takes the careful path).
__device__ __forceinline__ int count_eigs_below(const double (&m)[kMoments], double mu) {
    const double s00 = m[0] - mu, s01 = m[1], s02 = m[2], s11 = m[3] - mu, s12 = m[4], s22 = m[5] - mu;
    const double p0 = s00, r0 = rcp_fast(p0);
    const double l10 = s01 * r0, l20 = s02 * r0;
    const double p1 = fma(-l10, s01, s11), r1 = rcp_fast(p1);
    const double l21 = fma(-l20, s01, s12) * r1;
    const double p2 = fma(-l21 * l21, p1, fma(-l20, s02, s22)), r2 = rcp_fast(p2);
    int neg = 2 * ((p0 > 0.0 ? 0 : 1) + (p1 > 0.0 ? 0 : 1) + (p2 > 0.0 ? 0 : 1));
    double c00 = m[24] - mu, c01 = m[25], c02 = m[26], c11 = m[27] - mu, c12 = m[28], c22 = m[29] - mu;
#pragma unroll
    for (int b = 0; b < 2; ++b) {  // S1 = m[6..14], S2 = m[15..23], row-major 3x3
        double y[3][3], z[3][3];   // y = L^-1 S (forward substitution per column), z = D^-1 y
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double t0 = m[6 + 9 * b + j], t1 = m[9 + 9 * b + j], t2 = m[12 + 9 * b + j];
            y[0][j] = t0;
            y[1][j] = fma(-l10, t0, t1);
            y[2][j] = fma(-l21, y[1][j], fma(-l20, t0, t2));
            z[0][j] = y[0][j] * r0; z[1][j] = y[1][j] * r1; z[2][j] = y[2][j] * r2;
        }
#define APAP_SCHUR(i, j) fma(-y[2][i], z[2][j], fma(-y[1][i], z[1][j], -y[0][i] * z[0][j]))
        c00 += APAP_SCHUR(0, 0); c01 += APAP_SCHUR(0, 1); c02 += APAP_SCHUR(0, 2);
        c11 += APAP_SCHUR(1, 1); c12 += APAP_SCHUR(1, 2); c22 += APAP_SCHUR(2, 2);
#undef APAP_SCHUR
    }
    const double q0 = c00, k0 = rcp_fast(q0);
    const double g10 = c01 * k0, g20 = c02 * k0;
    const double q1 = fma(-g10, c01, c11), k1 = rcp_fast(q1);
    const double g21 = fma(-g20, c01, c12) * k1;
    const double q2 = fma(-g21 * g21, q1, fma(-g20, c02, c22));
    neg += (q0 > 0.0 ? 0 : 1) + (q1 > 0.0 ? 0 : 1) + (q2 > 0.0 ? 0 : 1);
    return neg;
}

// One Givens rotation of the row-insertion QR: eliminates r[K] against R[K][K] and updates the
// rest of both rows.  The hypotenuse is formed on operands scaled by a power of two, so weights
// down to the denormal range neither underflow in the squares nor lose the rotation's orthogonality.
template <int K>
__device__ __forceinline__ void givens_insert(double (&R)[45], double (&r)[9]) {
    const double a = R[tri(K, K)], b = r[K];
    const double big = fmax(fabs(a), fabs(b));
    // no rotation for an infinite entry (the scaling below would divide inf by inf).  A NaN entry is NOT filtered:
    // fmax ignores it, the rotation goes ahead and R turns NaN - like the reference's SVD of a matrix with a NaN
    // row (apap.py:159-161), whose cell is NaN too; such rows only come from non-finite keypoints.
    const bool rot = (b != 0.0) && (big < 1.797e308);
    const int e = rot ? __builtin_amdgcn_frexp_exp(big) : 0;
    const double sa = __builtin_ldexp(a, -e), sb = __builtin_ldexp(b, -e);
    const double t = sqrt(fma(sa, sa, sb * sb));
    const double c = rot ? sa / t : 1.0, s = rot ? sb / t : 0.0;
    R[tri(K, K)] = rot ? __builtin_ldexp(t, e) : a;
#pragma unroll
    for (int j = K + 1; j < 9; ++j) {
        const double x = R[tri(K, j)], y = r[j];
        R[tri(K, j)] = fma(c, x, s * y);
        r[j] = fma(c, y, -(s * x));
    }
    r[K] = 0.0;
}

template <int P, int Q>
__device__ __forceinline__ bool hestenes_rotate(double (&G)[81], double tol) {
    double alpha = 0.0, beta = 0.0, gam = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        alpha = fma(G[9 * i + P], G[9 * i + P], alpha);
        beta = fma(G[9 * i + Q], G[9 * i + Q], beta);
        gam = fma(G[9 * i + P], G[9 * i + Q], gam);
    }
    const bool rot = fabs(gam) > tol * sqrt(alpha * beta);  // false for gam == 0 and for NaN
    const double zeta = (beta - alpha) / (2.0 * gam);
    double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(fma(zeta, zeta, 1.0)));
    t = rot ? t : 0.0;
    const double c = 1.0 / sqrt(fma(t, t, 1.0));
    const double s = c * t;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const double gp = G[9 * i + P], gq = G[9 * i + Q];
        G[9 * i + P] = fma(c, gp, -(s * gq));
        G[9 * i + Q] = fma(s, gp, c * gq);
    }
    return rot;
}

// The careful path: what apap.py:159-161 computes, without squaring the condition number.
//   1. the weighted rows w_k * aa[2k], w_k * aa[2k+1] (float64 weight times float32 entry, one
//      rounding, like np.repeat(weight, 2)[:, None] * aa) are inserted one by one into a 9x9
//      upper-triangular R by Givens rotations (row-wise backward stable for any weight grading);
//      the float32 DLT entries are read back from the table, where they are stored exactly as
//      products with the row's constant 1;
//   2. one-sided Jacobi on the columns of G = R^T: G W has orthogonal columns sigma_j v_j, the right
//      singular vectors of W A - no accumulation of rotations is needed;
//   3. the column of the (pick_rank)-th smallest norm, normalised: the last row of the thin V^T.
// Lanes = cells; the wave walks all n keypoints (their table rows are wave-uniform scalar loads).
// Measured against a 60-digit SVD on the soak's ill-conditioned cells this is exact after the
// float32 rounding, and within 3e-5 px of numpy's float64 SVD (whose own error that is).
constexpr int kMaxHestenesSweeps = 30;

// `m24`: the rows come from a 24-sum table (apap_host_build_table24: x'x, x'y, y'x, y'y as float32 products in columns
// 24..27, -x' and -y' in columns 11 and 17).
__device__ __forceinline__ void qr_resolve(const double *__restrict__ table, int n, double vx, double vy,
                                        double inv_sigma, double gamma, int pick_rank, bool m24, double (&h)[9]) {
    double R[45];
#pragma unroll
    for (int k = 0; k < 45; ++k) R[k] = 0.0;
    for (int p = 0; p < n; ++p) {
        const double *__restrict__ row = table + (size_t)p * APAP_TABLE_STRIDE;
        const double w = cell_weight(vx, vy, row[30], row[31], inv_sigma, gamma);
        const double x = row[2], y = row[4];
        double r[9];
        r[0] = w * x; r[1] = w * y; r[2] = w; r[3] = 0.0; r[4] = 0.0; r[5] = 0.0;
        r[6] = w * row[m24 ? 24 : 12]; r[7] = w * row[m24 ? 25 : 13]; r[8] = w * row[m24 ? 11 : 14];
        givens_insert<0>(R, r); givens_insert<1>(R, r); givens_insert<2>(R, r);
        givens_insert<3>(R, r); givens_insert<4>(R, r); givens_insert<5>(R, r);
        givens_insert<6>(R, r); givens_insert<7>(R, r); givens_insert<8>(R, r);
        r[0] = 0.0; r[1] = 0.0; r[2] = 0.0; r[3] = w * x; r[4] = w * y; r[5] = w;
        r[6] = w * row[m24 ? 26 : 21]; r[7] = w * row[m24 ? 27 : 22]; r[8] = w * row[m24 ? 17 : 23];
        givens_insert<3>(R, r); givens_insert<4>(R, r); givens_insert<5>(R, r);
        givens_insert<6>(R, r); givens_insert<7>(R, r); givens_insert<8>(R, r);
    }
    // G = R^T scaled by a power of two so that squared column norms stay in range
    double big = 0.0;
#pragma unroll
    for (int k = 0; k < 45; ++k) big = fmax(big, fabs(R[k]));
    const int e = (big > 0.0 && big < 1.797e308) ? __builtin_amdgcn_frexp_exp(big) : 0;
    double G[81];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) G[9 * i + j] = (j <= i) ? __builtin_ldexp(R[tri(j, i)], -e) : 0.0;
    for (int sweep = 0; sweep < kMaxHestenesSweeps; ++sweep) {
        bool any = false;
#define APAP_PAIR(p, q) any |= hestenes_rotate<p, q>(G, 1e-15);
        APAP_PAIR(0, 1) APAP_PAIR(0, 2) APAP_PAIR(0, 3) APAP_PAIR(0, 4) APAP_PAIR(0, 5) APAP_PAIR(0, 6) APAP_PAIR(0, 7) APAP_PAIR(0, 8)
        APAP_PAIR(1, 2) APAP_PAIR(1, 3) APAP_PAIR(1, 4) APAP_PAIR(1, 5) APAP_PAIR(1, 6) APAP_PAIR(1, 7) APAP_PAIR(1, 8)
        APAP_PAIR(2, 3) APAP_PAIR(2, 4) APAP_PAIR(2, 5) APAP_PAIR(2, 6) APAP_PAIR(2, 7) APAP_PAIR(2, 8)
        APAP_PAIR(3, 4) APAP_PAIR(3, 5) APAP_PAIR(3, 6) APAP_PAIR(3, 7) APAP_PAIR(3, 8)
        APAP_PAIR(4, 5) APAP_PAIR(4, 6) APAP_PAIR(4, 7) APAP_PAIR(4, 8)
        APAP_PAIR(5, 6) APAP_PAIR(5, 7) APAP_PAIR(5, 8)
        APAP_PAIR(6, 7) APAP_PAIR(6, 8)
        APAP_PAIR(7, 8)
#undef APAP_PAIR
        if (!__any(any)) break;
    }
    double nrm2[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) acc = fma(G[9 * i + j], G[9 * i + j], acc);
        nrm2[j] = acc;
    }
    // singular values sorted descending, equal ones in index order (a zero matrix gives V = I):
    // rank = number of columns that sort AFTER this one
    int best = 8;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        int rank = 0;
#pragma unroll
        for (int j = 0; j < 9; ++j)
            if (j != i) rank += (nrm2[j] < nrm2[i] || (nrm2[j] == nrm2[i] && j > i)) ? 1 : 0;
        best = (rank == pick_rank) ? i : best;
    }
    double nb = nrm2[0];
#pragma unroll
    for (int c = 1; c < 9; ++c) nb = (best == c) ? nrm2[c] : nb;
    const double inv = nb > 0.0 ? 1.0 / sqrt(nb) : 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        double x = G[9 * k];
#pragma unroll
        for (int c = 1; c < 9; ++c) x = (best == c) ? G[9 * k + c] : x;
        h[k] = nb > 0.0 ? x * inv : (k == best ? 1.0 : 0.0);
    }
}

// What the solve's tail needs to leave a cell "warp ready": where the warp's per-cell tables of this pair live and the
// geometry the float32 estimate's anchors come from.  hinv_pad == nullptr: the solve writes H only.
struct WarpEmit {
    double *hinv_pad;       // pair 0's tables (apap_warp_batch_device's workspace); pair k at + k * stride
    float4 *frec;
    long long hinv_stride, frec_stride;
    const double *mesh_w, *mesh_h;
    int n_w, n_h, mesh_rows, mesh_cols, final_w, final_h, off_x, off_y;
    int *status;
};
// the four mesh edges around a cell, fetched by the solve's kernels BEFORE the eigen-solve (one wave per SIMD there: a load at
// the very end of the tail would be a fully exposed round trip)
struct CellEdges {
    double w0, w1, h0, h1;
};
__device__ __forceinline__ CellEdges warp_emit_prefetch(const WarpEmit &we, int cell) {
    CellEdges e = {0.0, 0.0, 0.0, 0.0};
    if (we.hinv_pad && cell >= 0) {
        const int c = min(cell, we.mesh_rows * we.mesh_cols - 1);
        const int er = c / we.mesh_cols, ec = c - er * we.mesh_cols;
        // (the same guards as fast_origin: a mesh with fewer edges than cells + 1 has no ordinary cells)
        if (ec + 1 < we.n_w) { e.w0 = we.mesh_w[ec]; e.w1 = we.mesh_w[ec + 1]; }
        if (er + 1 < we.n_h) { e.h0 = we.mesh_h[er]; e.h1 = we.mesh_h[er + 1]; }
    }
    return e;
}
__device__ void warp_emit_from_solve(const WarpEmit &we, int pair, int cell, const float (&Hf)[9], const CellEdges &edges);

// The per-cell tail shared by K2 and the fused small-mesh kernel: from the 30 moment sums of a
// cell to its float32 homography.  Lanes = cells; inactive lanes are simply not in the wave votes.
template <bool kUseInverseIteration>
__device__ __forceinline__ void eigen_denorm_cell(const double (&m)[kMoments], const double *__restrict__ denorm,
                                                  int pick_rank, int careful, const double *__restrict__ table, int n,
                                                  double vx, double vy, double gamma, double inv_sigma,
                                                  float *__restrict__ out /* 9 floats, or nullptr */,
                                                  const WarpEmit &we, int pair, int cell, const CellEdges &edges, bool m24 = false) {
    // A^T W^2 A = [[S0, 0, S1], [0, S0, S2], [S1^T, S2^T, S3]]  (3x3 blocks)
    double a[45];
#pragma unroll
    for (int k = 0; k < 45; ++k) a[k] = 0.0;
    a[tri(0, 0)] = m[0]; a[tri(0, 1)] = m[1]; a[tri(0, 2)] = m[2];
    a[tri(1, 1)] = m[3]; a[tri(1, 2)] = m[4]; a[tri(2, 2)] = m[5];
    a[tri(3, 3)] = m[0]; a[tri(3, 4)] = m[1]; a[tri(3, 5)] = m[2];
    a[tri(4, 4)] = m[3]; a[tri(4, 5)] = m[4]; a[tri(5, 5)] = m[5];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            a[tri(i, 6 + j)] = m[6 + 3 * i + j];
            a[tri(3 + i, 6 + j)] = m[15 + 3 * i + j];
        }
    a[tri(6, 6)] = m[24]; a[tri(6, 7)] = m[25]; a[tri(6, 8)] = m[26];
    a[tri(7, 7)] = m[27]; a[tri(7, 8)] = m[28]; a[tri(8, 8)] = m[29];
    const double trace = 2.0 * (m[0] + m[3] + m[5]) + m[24] + m[27] + m[29];
    const double gap_tol = kGapTol * trace;
    // The sums are of w^2: where every weight is below ~1e-140 they underflow (partly or entirely)
    // although the weights themselves - what the reference's SVD works on - are ordinary numbers.
    const bool underflowed = !(trace >= 1e-280) || !(trace < 1.797e308);   // also NaN / inf sums

    double h[9];
    bool have = false;
    // Fewer than 5 keypoints (pick_rank != 0): the reference's V[-1] is the singular vector of the
    // smallest KEPT singular value of the thin SVD, lambda_(9-2n) of a matrix with 9 - 2n exact
    // zeros below it: always the careful path.
    bool careful_cell = careful && (pick_rank != 0 || underflowed);
    if (kUseInverseIteration && pick_rank == 0) {
        double rho;
        int below;
        have = inverse_iteration(a, h, rho, m, gap_tol, below);
        // a second eigenvalue within gap_tol of the smallest?  (rho >= lambda_9, so lambda_9 counts)
        if (careful && !careful_cell) careful_cell = have && below >= 2;
    }
    if (!__all(have || careful_cell)) {  // rare: no spectral gap, pivot not positive (n < 5 with careful == 0)
        double hj[9];
        const double gap = jacobi_eigvec(a, pick_rank, hj);
        if (careful && pick_rank == 0 && !have && !careful_cell) careful_cell = !(gap >= gap_tol);  // also for NaN
#pragma unroll
        for (int k = 0; k < 9; ++k) h[k] = have ? h[k] : hj[k];
    }
    if (__any(careful_cell)) {
        double hq[9];
        qr_resolve(table, n, vx, vy, inv_sigma, gamma, pick_rank, m24, hq);
#pragma unroll
        for (int k = 0; k < 9; ++k) h[k] = careful_cell ? hq[k] : h[k];
    }
    double t1[9], t2[9];
    mul3(denorm, h, t1);        // inv(C2) . h
    mul3(t1, denorm + 9, t2);   // . C1
    mul3(denorm + 18, t2, t1);  // inv(N2) .
    mul3(t1, denorm + 27, t2);  // . N1
    if (out) {
        // h / h[2,2] (apap.py:167): one reciprocal and a residual correction per quotient (the Markstein
        // sequence of k_warp_rows: the correctly rounded quotient) instead of nine IEEE division sequences;
        // anything out of the ordinary (h[2,2] zero / tiny / huge / NaN, an overflowing quotient) takes those.
        const double r8 = rcp_full(t2[8]);
        double q[9];
        bool plain = fabs(t2[8]) >= 1e-290 && fabs(t2[8]) <= 1e290;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double q0 = t2[k] * r8;
            q[k] = fma(fma(-t2[8], q0, t2[k]), r8, q0);
            plain = plain && fabs(q[k]) <= 1e290;
        }
        q[8] = 1.0;
        if (!__all(plain)) {
#pragma unroll
            for (int k = 0; k < 9; ++k) q[k] = plain ? q[k] : t2[k] / t2[8];
        }
        float hf[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) out[k] = hf[k] = (float)q[k];
        // the caller warps next: the cell's inverse, its float32-estimate record and exact-path floats, straight into the
        // warp's workspace (what k_warp_setup's per-cell half would compute from the stored grid: the same function)
        if (we.hinv_pad) warp_emit_from_solve(we, pair, cell, hf, edges);
    }
}

// K2.  kUseInverseIteration = false is the pure-Jacobi kernel (APAP_EIGEN_JACOBI).
// kM24: the slabs hold the 24 sums S0, Sx, Sy, Sr (6 each, upper triangle of a 3 x 3 by rows) of K1's 24-sum form; the 30
// entries the tail works on are those with Sx and Sy written out as full 3 x 3 blocks.
template <bool kUseInverseIteration, bool kM24>
__global__ __launch_bounds__(64) void k_eigen_denorm(const double *__restrict__ moments, int splits,
                                                     int cells, int cells_pad,
                                                     const double *__restrict__ denorm, int pick_rank,
                                                     float *__restrict__ H, BatchStride bs,
                                                     const double *__restrict__ table, int n,
                                                     const double *__restrict__ vertices, double gamma,
                                                     double inv_sigma, int careful, const WarpEmit we) {
    moments += (long long)blockIdx.z * bs.moments;
    denorm += (long long)blockIdx.z * bs.denorm;
    H += (long long)blockIdx.z * bs.H;
    table += (long long)blockIdx.z * bs.table;
    vertices += (long long)blockIdx.z * bs.vertices;
    const int cell = blockIdx.x * kWave + threadIdx.x;
    const int cc = min(cell, cells - 1);
    const CellEdges edges = warp_emit_prefetch(we, cell < cells ? cell : -1);
    constexpr int kSums = kM24 ? 24 : kMoments;
    const bool layout_ok = table_is_24(table) == kM24;   // (one scalar load, long before the tail needs it)
    double t[kSums];
#pragma unroll
    for (int j = 0; j < kSums; ++j) t[j] = moments[(size_t)j * cells_pad + cc];
    for (int g = 1; g < splits; ++g) {  // slabs of the keypoint splits, fixed order
        const double *slab = moments + (size_t)g * kSums * cells_pad + cc;
#pragma unroll
        for (int j = 0; j < kSums; ++j) t[j] += slab[(size_t)j * cells_pad];
    }
    double m[kMoments];
    if constexpr (kM24) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            m[j] = t[j];
            m[24 + j] = t[18 + j];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int u = i <= j ? i * (5 - i) / 2 + j : j * (5 - j) / 2 + i;   // (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
                m[6 + 3 * i + j] = t[6 + u];
                m[15 + 3 * i + j] = t[12 + u];
            }
    } else {
#pragma unroll
        for (int j = 0; j < kMoments; ++j) m[j] = t[j];
    }
    float *out = cell < cells ? H + (size_t)cell * 9 : nullptr;
    eigen_denorm_cell<kUseInverseIteration>(m, denorm, pick_rank, careful, table, n, vertices[2 * cc], vertices[2 * cc + 1],
                                            gamma, inv_sigma, out, we, (int)blockIdx.z, cell, edges, kM24);
    if (!layout_ok && out) {      // the caller's table has the other layout (wave-uniform): nothing computed from it is an answer
#pragma unroll
        for (int k = 0; k < 9; ++k) out[k] = __builtin_nanf("");
    }
}

// --------------------------------------------------------------------------------
// K1 + K2 fused for small meshes (a few thousand cells: config C1 is 400).  There the two-launch
// path is bound by launch latency and by the few waves it can start, not by arithmetic.  Here a
// block owns ONE 16-cell group; its four waves split the keypoints (wave w takes steps w, w+4,
// w+8, w+12 of every 64-keypoint LDS chunk), add their accumulators through LDS in a fixed order,
// and wave 0 finishes the 16 cells with the K2 tail: one launch, no moment slabs in HBM, four times
// as many blocks as K1 would start.  Register use is the K2 tail's (one wave per SIMD): fine for the
// few dozen to few hundred blocks this kernel is dispatched for.
// --------------------------------------------------------------------------------
constexpr int kSmallWaves = 4;       // waves of a fused block: they split every chunk's 16 steps
constexpr int kSmallThreads = kSmallWaves * 64;
static_assert(kSmallWaves == 4 || kSmallWaves == 8, "the chunk's 16 steps and the staging split evenly over 4 or 8 waves");
template <bool kUseInverseIteration>
__global__ __launch_bounds__(kSmallThreads) void k_solve_small(const double *__restrict__ table, int n,
                                                     const double *__restrict__ vertices, int cells, double gamma,
                                                     double inv_sigma, const double *__restrict__ denorm, int pick_rank,
                                                     int careful, float *__restrict__ H, BatchStride bs, const WarpEmit we) {
    // The block is alone on its CU (the K2 tail's registers) and its waves take only 4 steps of a
    // chunk each, far too little work to hide the next chunk's global load behind: kRing chunks are
    // fetched per round with their loads all in flight together (C1's 150 keypoints: one round, one
    // exposed latency), and the next round's loads fly during this round's arithmetic.
    constexpr int kRing = 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[kRing][kChunk * 256];
    __shared__ double s_exp2[kExpN];
    table += (long long)blockIdx.z * bs.table;
    vertices += (long long)blockIdx.z * bs.vertices;
    denorm += (long long)blockIdx.z * bs.denorm;
    H += (long long)blockIdx.z * bs.H;
    const int tid = threadIdx.x;
    constexpr int kExpPerThread = kExpN / kSmallThreads;
    double exp_stage[kExpPerThread];   // to LDS after the first round's loads are issued
#pragma unroll
    for (int i = 0; i < kExpPerThread; ++i) exp_stage[i] = kExp2Tab[tid + i * kSmallThreads];
    // the four de-normalisation matrices too: the tail would otherwise wait for them (cold scalar loads) at its very end
    __shared__ double s_denorm[APAP_DENORM_DOUBLES];
    const double den_stage = denorm[min(tid, APAP_DENORM_DOUBLES - 1)];
    static_assert(kExpN % kSmallThreads == 0, "exp table entries per thread");
    const double gamma2 = gamma > 0.0 ? gamma * gamma : 0.0;
    const double scaled_inv_sigma2 = 2.0 * inv_sigma * kExpScale;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kgrp = lane >> 4, col = lane & 15;
    const int cell = blockIdx.x * 16 + col;
    const int cc = min(cell, cells - 1);
    const double vx = vertices[2 * cc], vy = vertices[2 * cc + 1];
    const int nchunks = (n + kChunk - 1) / kChunk;
    const int nrounds = (nchunks + kRing - 1) / kRing;

    constexpr int kPieces = kChunk * 16 / kSmallThreads;
    double2 stage[kRing][kPieces];
    auto load_round = [&](int r) {
#pragma unroll
        for (int b = 0; b < kRing; ++b) {
#pragma unroll
            for (int i = 0; i < kPieces; ++i) {
                const int q = tid + kSmallThreads * i;
                const int p = (r * kRing + b) * kChunk + (q >> 4);
                stage[b][i] = (p < n) ? *reinterpret_cast<const double2 *>(table + (size_t)p * APAP_TABLE_STRIDE + 2 * (q & 15))
                                      : make_double2(0.0, 0.0);  // zero rows add nothing whatever their weight
            }
        }
    };
    auto store_round = [&]() {
#pragma unroll
        for (int b = 0; b < kRing; ++b) {
#pragma unroll
            for (int i = 0; i < kPieces; ++i) {
                const int q = tid + kSmallThreads * i;
                const int r = q >> 4;
                const int slot = (q & 15) ^ ((r & 1) << 3);
                *reinterpret_cast<double2 *>(&lds[b][r * 256 + slot * 16]) = stage[b][i];
            }
        }
    };
    const int off_xy = lds_off(kgrp, 30), off_b0 = lds_off(kgrp, col), off_b1 = lds_off(kgrp, 16 + col);
    double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    load_round(0);
#pragma unroll
    for (int i = 0; i < kExpPerThread; ++i) s_exp2[tid + i * kSmallThreads] = exp_stage[i];
    if (tid < APAP_DENORM_DOUBLES) s_denorm[tid] = den_stage;
    store_round();
    __syncthreads();
    for (int r = 0; r < nrounds; ++r) {
        if (r + 1 < nrounds) load_round(r + 1);
        const int live = min(kRing, nchunks - r * kRing);   // chunks of this round (block-uniform)
        for (int b = 0; b < live; ++b) {
            const unsigned char *buf = lds[b] + 1024 * wave;
            // only the steps that hold keypoints (see k_assemble_mfma): C1's 150 keypoints are 2 chunks and 6 steps
            const int steps_here = min(kChunk / 4, (n - (r * kRing + b) * kChunk + 3) >> 2);
            if (steps_here < kChunk / 4) {
                for (int i = 0; wave + kSmallWaves * i < steps_here; ++i) {
                    const double2 xy = *reinterpret_cast<const double2 *>(buf + off_xy + 1024 * kSmallWaves * i);
                    const double b0 = *reinterpret_cast<const double *>(buf + off_b0 + 1024 * kSmallWaves * i);
                    const double b1 = *reinterpret_cast<const double *>(buf + off_b1 + 1024 * kSmallWaves * i);
                    const double w2s = cell_weight_sq_tab(vx, vy, xy.x, xy.y, scaled_inv_sigma2, gamma2, s_exp2);
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w2s, b0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w2s, b1, acc1, 0, 0, 0);
                }
                continue;
            }
            constexpr int kMine = kChunk / 4 / kSmallWaves;
            double w2[kMine];  // this wave's steps: wave, wave + kSmallWaves, ...; weights first (see k_assemble_mfma)
#pragma unroll
            for (int i = 0; i < kMine; ++i) {
                const double2 xy = *reinterpret_cast<const double2 *>(buf + off_xy + 1024 * kSmallWaves * i);
                w2[i] = cell_weight_sq_tab(vx, vy, xy.x, xy.y, scaled_inv_sigma2, gamma2, s_exp2);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < kMine; ++i) {
                const double b0 = *reinterpret_cast<const double *>(buf + off_b0 + 1024 * kSmallWaves * i);
                const double b1 = *reinterpret_cast<const double *>(buf + off_b1 + 1024 * kSmallWaves * i);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w2[i], b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w2[i], b1, acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                    // every wave is done reading the ring
        if (r + 1 < nrounds) {
            store_round();
            __syncthreads();
        }
    }
    // cross-wave reduction through LDS (the chunk buffers are free now): part[wave][cell][32 moments], rows padded
    // to 33 doubles so that both "32 moments of a cell" and "one moment of 16 cells" are conflict-free (with 32, the
    // tail's 16 lanes read 256 B apart: one bank, 16 turns per read).  All threads add the waves' partial sums (in
    // wave order: the same sums whoever adds them); the tail only picks up its cell's 30 totals.
    // D layout: register i of lane l is D[row = (l >> 4) + 4 i][col = l & 15] = (cell, moment)
    constexpr int kRow = 33;
    double *part = reinterpret_cast<double *>(&lds[0][0]);
    double *tot = part + kSmallWaves * 16 * kRow;
    static_assert((kSmallWaves + 1) * 16 * kRow * sizeof(double) <= sizeof(lds), "reduction buffers fit the chunk ring");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ci = kgrp + 4 * i;
        part[(wave * 16 + ci) * kRow + col] = acc0[i];
        part[(wave * 16 + ci) * kRow + 16 + col] = acc1[i];
    }
    __syncthreads();
#pragma unroll
    for (int e = tid; e < 16 * 32; e += kSmallThreads) {
        const int ce = e >> 5, mo = e & 31;
        double sum = part[(0 * 16 + ce) * kRow + mo];
#pragma unroll
        for (int w = 1; w < kSmallWaves; ++w) sum += part[(w * 16 + ce) * kRow + mo];
        tot[ce * kRow + mo] = sum;
    }
    __syncthreads();
    if (wave != 0 || lane >= 16) return;       // lanes 0..15 of wave 0 = the block's 16 cells
    double m[kMoments];
#pragma unroll
    for (int j = 0; j < kMoments; ++j) m[j] = tot[lane * kRow + j];
    const CellEdges edges = warp_emit_prefetch(we, cell < cells ? cell : -1);
    eigen_denorm_cell<kUseInverseIteration>(m, s_denorm, pick_rank, careful, table, n, vx, vy, gamma, inv_sigma,
                                            cell < cells ? H + (size_t)cell * 9 : nullptr, we, (int)blockIdx.z, cell, edges);
    if (table_is_24(table) && cell < cells) {   // a 24-sum table: this kernel reads the 30-sum layout
#pragma unroll
        for (int k = 0; k < 9; ++k) H[(size_t)cell * 9 + k] = __builtin_nanf("");
    }
}

// --------------------------------------------------------------------------------
// optional second return value of local_homography: the (cells, n) weight tensor.
// HBM-write-bound (8 n bytes per cell).
// --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_weights(const double *__restrict__ table, int n,
                                                 const double *__restrict__ vertices, int cells,
                                                 double gamma, double inv_sigma, double *__restrict__ W) {
    const size_t total = (size_t)cells * n;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int cell = (int)(idx / n);
        const int p = (int)(idx - (size_t)cell * n);
        const double *row = table + (size_t)p * APAP_TABLE_STRIDE;
        W[idx] = cell_weight(vertices[2 * cell], vertices[2 * cell + 1], row[30], row[31], inv_sigma, gamma);
    }
}

// --------------------------------------------------------------------------------
// 3x3 inverse in float64 by LU with partial pivoting (what numpy.linalg.inv does for a
// float32 input: dgesv on the widened matrix, result cast back - apap.py:203,252).
// Branch-free row swaps so that everything stays in registers.  Returns false on an
// exactly-zero pivot.
// --------------------------------------------------------------------------------
__device__ __forceinline__ void cswap(bool c, double &x, double &y) {
    const double tx = c ? y : x, ty = c ? x : y;
    x = tx;
    y = ty;
}

// a / b with 1 / b in hand (rb = rcp_full(b)): the product and one residual correction - the Markstein sequence, which
// returns the correctly rounded quotient except in a vanishing share of cases (then 1 ulp off); `plain` false (a pivot that
// is zero, denormal, huge or NaN): the IEEE division.  Twelve of these replace twelve ~30-instruction division sequences in
// inv3, which runs once per cell in the solve's tail and in the warp's set-up.
template <bool kFast>
__device__ __forceinline__ double quot(double a, double b, double rb) {
    if (!kFast) return a / b;
    const double q0 = a * rb;
    return fma(fma(-b, q0, a), rb, q0);
}

// `plain` (kFast only): every pivot was an ordinary number - the reciprocal-based quotients are valid
template <bool kFast>
__device__ __forceinline__ bool inv3_impl(const double *m, double *out, bool &plain) {
    double a00 = m[0], a01 = m[1], a02 = m[2], a10 = m[3], a11 = m[4], a12 = m[5], a20 = m[6], a21 = m[7], a22 = m[8];
    double b00 = 1, b01 = 0, b02 = 0, b10 = 0, b11 = 1, b12 = 0, b20 = 0, b21 = 0, b22 = 1;
    bool ok = true, p0, p1;
    double r0, r1;
    // column 0 pivot: first maximal |a_i0|
    {
        const bool s1 = fabs(a10) > fabs(a00);
        cswap(s1, a00, a10); cswap(s1, a01, a11); cswap(s1, a02, a12);
        cswap(s1, b00, b10); cswap(s1, b01, b11); cswap(s1, b02, b12);
        const bool s2 = fabs(a20) > fabs(a00);
        cswap(s2, a00, a20); cswap(s2, a01, a21); cswap(s2, a02, a22);
        cswap(s2, b00, b20); cswap(s2, b01, b21); cswap(s2, b02, b22);
        // the two conditional swaps above pick the max but may permute the two
        // non-pivot rows differently from LAPACK; the inverse does not depend on it
        // beyond rounding at the 1e-16 level.
        ok = ok && (a00 != 0.0);
        p0 = fabs(a00) >= 1e-290 && fabs(a00) <= 1e290;
        r0 = kFast ? rcp_full(a00) : 0.0;
        const double l1 = quot<kFast>(a10, a00, r0), l2 = quot<kFast>(a20, a00, r0);
        a11 = fma(-l1, a01, a11); a12 = fma(-l1, a02, a12);
        b10 = fma(-l1, b00, b10); b11 = fma(-l1, b01, b11); b12 = fma(-l1, b02, b12);
        a21 = fma(-l2, a01, a21); a22 = fma(-l2, a02, a22);
        b20 = fma(-l2, b00, b20); b21 = fma(-l2, b01, b21); b22 = fma(-l2, b02, b22);
    }
    {
        const bool s = fabs(a21) > fabs(a11);
        cswap(s, a11, a21); cswap(s, a12, a22);
        cswap(s, b10, b20); cswap(s, b11, b21); cswap(s, b12, b22);
        ok = ok && (a11 != 0.0);
        p1 = fabs(a11) >= 1e-290 && fabs(a11) <= 1e290;
        r1 = kFast ? rcp_full(a11) : 0.0;
        const double l = quot<kFast>(a21, a11, r1);
        a22 = fma(-l, a12, a22);
        b20 = fma(-l, b10, b20); b21 = fma(-l, b11, b21); b22 = fma(-l, b12, b22);
    }
    ok = ok && (a22 != 0.0);
    const bool p2 = fabs(a22) >= 1e-290 && fabs(a22) <= 1e290;
    const double r2 = kFast ? rcp_full(a22) : 0.0;
    plain = p0 && p1 && p2;
    // back substitution, column by column
    const double x20 = quot<kFast>(b20, a22, r2), x21 = quot<kFast>(b21, a22, r2), x22 = quot<kFast>(b22, a22, r2);
    const double x10 = quot<kFast>(fma(-a12, x20, b10), a11, r1), x11 = quot<kFast>(fma(-a12, x21, b11), a11, r1),
                 x12 = quot<kFast>(fma(-a12, x22, b12), a11, r1);
    const double x00 = quot<kFast>(fma(-a02, x20, fma(-a01, x10, b00)), a00, r0);
    const double x01 = quot<kFast>(fma(-a02, x21, fma(-a01, x11, b01)), a00, r0);
    const double x02 = quot<kFast>(fma(-a02, x22, fma(-a01, x12, b02)), a00, r0);
    out[0] = x00; out[1] = x01; out[2] = x02;
    out[3] = x10; out[4] = x11; out[5] = x12;
    out[6] = x20; out[7] = x21; out[8] = x22;
    return ok;
}

// 3 x 3 inverse by LU with partial pivoting (what numpy.linalg.inv does in float64, apap.py:203).  The common path takes its
// twelve quotients from three reciprocals; a wave that holds a cell with a zero, denormal, huge or NaN pivot redoes its cells
// with IEEE divisions (wave-uniform branch).
__device__ __forceinline__ bool inv3(const double *m, double *out) {
    bool plain;
    bool ok = inv3_impl<true>(m, out, plain);
    if (!__all(plain)) {
        double o2[9];
        bool dummy;
        const bool ok2 = inv3_impl<false>(m, o2, dummy);
#pragma unroll
        for (int k = 0; k < 9; ++k) out[k] = plain ? out[k] : o2[k];
        ok = plain ? ok : ok2;
    }
    return ok;
}

// per-cell inverse for the warp (apap.py:201-203): padded float32 copy for the warp
// kernel, optional dense copy for the caller (the reference's mutated argument).
template <typename T>   // T = float: the reference's float32 grid; double: a float64 grid stays float64 (apap.py:201-203)
__global__ __launch_bounds__(256) void k_invert_cells(const T *__restrict__ H, int cells,
                                                      double *__restrict__ hinv_pad,
                                                      T *__restrict__ hinv_dense, int *status) {
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= cells) return;
    double m[9], r[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) m[k] = (double)H[(size_t)cell * 9 + k];
    if (!inv3(m, r)) atomicOr(status, apap::kStatusSingular);
    double *p = hinv_pad + (size_t)cell * APAP_HINV_STRIDE;
#pragma unroll
    for (int k = 0; k < 9; ++k) p[k] = (double)(T)r[k];  // the inverse in the grid's own dtype, widened once
    p[9] = 0.0;
    if (hinv_dense) {
#pragma unroll
        for (int k = 0; k < 9; ++k) hinv_dense[(size_t)cell * 9 + k] = (T)r[k];
    }
}

// canvas row -> cell row and canvas column -> cell column:
// "first k with i < edges[k]" minus one (apap.py:207,209-210), -1 wrapping to the last
// cell like a Python index.  No monotonicity is assumed, hence the linear scan.
// The lookup tables of a warp workspace (lut, fcol, frow) hold CELL INDICES: a gather that ran on a workspace whose tables were
// never built - or were built for another mesh / canvas size - would index the per-cell tables with garbage.  The word behind
// the lut carries a stamp of the sizes the tables were built for; every gather kernel compares it (one scalar load, issued with
// its first table loads) before it uses a table VALUE as an index, and leaves with bit 2 of the status word otherwise.
__host__ __device__ __forceinline__ int warp_stamp(int mesh_rows, int mesh_cols, int final_w, int final_h) {
    return (int)(((unsigned)mesh_rows * 0x9E3779B1u) ^ ((unsigned)mesh_cols * 0x85EBCA77u) ^ ((unsigned)final_w * 0xC2B2AE3Du) ^
                 ((unsigned)final_h * 0x27D4EB2Fu)) | 1;
}
// the same in 16 bits, never 0: carried by every row entry of the fast tables (frow[i].x = cell row | stamp16 << 16), so that
// the strip kernel learns it from a load it makes anyway
__host__ __device__ __forceinline__ unsigned warp_stamp16(int mesh_rows, int mesh_cols, int final_w, int final_h) {
    return (((unsigned)warp_stamp(mesh_rows, mesh_cols, final_w, final_h) >> 12) & 0xffffu) | 1u;
}
__device__ __forceinline__ bool warp_tables_ready(int have, int mesh_rows, int mesh_cols, int final_w, int final_h, int *status) {
    if (__builtin_expect(have == warp_stamp(mesh_rows, mesh_cols, final_w, final_h), 1)) return true;
    if ((threadIdx.x & 63) == 0) atomicOr(status, apap::kStatusUnprepared);
    return false;
}

__global__ __launch_bounds__(256) void k_cell_lut(const double *__restrict__ mesh_w, int n_w,
                                                  const double *__restrict__ mesh_h, int n_h,
                                                  int mesh_rows, int mesh_cols, int final_w,
                                                  int final_h, int *__restrict__ lut, int *status) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) lut[final_w + final_h] = warp_stamp(mesh_rows, mesh_cols, final_w, final_h);
    if (t >= final_h + final_w) return;
    const bool is_row = t < final_h;
    const int idx = is_row ? t : t - final_h;
    const double *edges = is_row ? mesh_h : mesh_w;
    const int n_e = is_row ? n_h : n_w;
    const int ncell = is_row ? mesh_rows : mesh_cols;
    int first = -1;
    const double x = (double)idx;
    for (int k = 0; k < n_e; ++k) {
        if (x < edges[k]) { first = k; break; }
    }
    int c = first - 1;
    bool bad = first < 0;
    if (c < 0) c += ncell;
    bad = bad || c < 0 || c >= ncell;
    if (bad) { atomicOr(status, apap::kStatusIndex); c = 0; }
    lut[t] = c;
}

// Warp set-up in ONE launch: blocks [0, inv_blocks) invert 256 cells each; the following
// blocks build the lookup table, 1024 canvas rows (then columns) per block.
// "first k with i < edges[k]" is the first k whose RUNNING MAXIMUM exceeds i (an edge
// that is not above i cannot be above anything a later comparison still needs), and the
// running maximum is monotone: an inclusive max-scan of the edges in LDS, then a binary
// search per index.  Valid for any edge order, like the reference's np.where scan.
constexpr int kMaxEdges = 4096;  // per axis; larger meshes take the linear-scan kernel

// ---- K3 fast path: float32 estimate + doubt window (k_warp_fast below) -------------------------
// What decides a canvas pixel is int(tx), int(ty) and the strict test 0 < t < size (apap.py:211-215),
// not the bits of the float64 coordinates.  k_warp_fast therefore works on a float32 ESTIMATE of the
// coordinates relative to an integer anchor and takes the exact float64 sequence (target_from) only
// for pixels whose estimate lies within a proven error bound of an integer.
//
// Per mesh cell the set-up kernel picks an anchor pixel (xb, yb) = the cell's first canvas column and
// row, and with h = the cell's stored inverse, t_i(x, y) = h[3i] x + h[3i+1] y + h[3i+2] (real
// arithmetic), Q = fl(t0(xb, yb) / t2(xb, yb)), n0 = floor(Q), f0 = Q - n0:
//     tx - n0 = f0 + (R + A dx + B dy) / t2(xb + dx, yb + dy),   A = h0 - Q h6,  B = h1 - Q h7,
// R = t0b - Q t2b a rounding residual.  The kernel evaluates, in float32 with exactly two FMAs per
// sum, one v_rcp_f32 and one multiply,
//     s = (c' + a' dx + b' dy) / (t2b + h6 dx + h7 dy),   a' = A + f h6, b' = B + f h7, c' = f t2b,
// with f = f0 + dE: the estimate of tx - n0 shifted up by dE, everything scaled by 2^16 so that
// F = floor(s) (v_cvt_flr_i32_f32) is a 16.16 fixed-point number whose integer part and fraction are the two
// 16-bit halves of the register - the anchor add and the doubt test take them through SDWA operand selects,
// no shift instruction (round 5: 17 instead of 21 vector instructions per pixel; the 10.22 format of rounds 3-4
// paid two shifts for each).  With E >= |error of s| + |tx_ref - tx| (bounded below from the magnitudes of this
// cell; tx_ref = what the reference's float64 sequence returns) and dE strictly above E:  v = tx_ref - n0 lies in
// (s u - 2 dE, s u), u = 2^-16, and s in [F, F + 1), so if the low 16 bits of F are >= ceil(2 dE / u), then
// n0 + (F >> 16) < tx_ref < n0 + (F >> 16) + 1 strictly - the truncation, and both strict range tests, follow
// from the integer alone.  Otherwise the pixel is "in doubt" and is recomputed exactly (the window is at least one
// unit of 2^-16 px wide: 3e-5 of the pixels of C2 / C3, oracle/warp_fast_spec.py).
// Cells where the bound cannot be established (perspective denominators that change sign or
// by more than a quarter across the cell, coordinates beyond 2^30, cells wider than 254 pixels, mesh
// edges that are not increasing) get a record whose window covers everything: correctness never
// depends on the estimate.
constexpr int kFastFracBits = 16;    // 16.16: integer part and fraction are the two halves of the dword (SDWA operand selects)
constexpr int kFastMaxSpan = 254;   // dx, dy travel as bytes

// anchor of cell index c along one axis: first pixel at or after its lower edge, and how many pixels
// the cell spans (clamped to kFastMaxSpan); false when the edges do not describe an ordinary cell
__device__ __forceinline__ bool fast_origin_of(double e0, double e1, bool have, int count, int &x0, int &span);
__device__ __forceinline__ bool fast_origin(const double *__restrict__ edges, int n_e, int c, int count, int &x0, int &span) {
    x0 = 0;
    span = 1;
    if (c < 0 || c + 1 >= n_e) return false;
    return fast_origin_of(edges[c], edges[c + 1], true, count, x0, span);
}
__device__ __forceinline__ bool fast_origin_of(double e0, double e1, bool have, int count, int &x0, int &span) {
    x0 = 0;
    span = 1;
    if (!have) return false;
    if (!(e0 > -1.0) || !(e1 > e0) || !(e0 < 2147483000.0)) return false;   // NaN fails every test
    const double a = fmin(fmax(ceil(e0), 0.0), (double)count);
    const double b = fmin(fmax(ceil(e1), 0.0), (double)count);
    if (!(b > a)) return false;          // no pixel of the canvas lies in this cell
    x0 = (int)a;
    span = (int)fmin(b - a, (double)kFastMaxSpan);
    return true;
}

// the 12-dword record k_warp_fast reads per cell: {a'x, b'x, c'x, a'y | b'y, c'y, t2b, h6 | h7, n0x, n0y, thr}
// `src_rows` (optional): lower / upper bound of the SOURCE rows the cell's pixels can read - n0y -+ a bound of
// |estimate| - merged into two ints with atomicMin / atomicMax; a cell without a bound claims every row.
__device__ __forceinline__ void fast_record(const double (&h)[9], bool origin_ok, double xb, double yb, double DX,
                                            double DY, float4 *__restrict__ out, int *src_rows = nullptr) {
    const double eps64 = 1.1102230246251565e-16, eps32 = 5.9604644775390625e-08, unit = 65536.0;   // 2^16
    float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa, pc = pa;
    pc.w = __uint_as_float(0xffffffffu);   // window = everything: den = 0 -> s = NaN -> int(s) = 0 < thr
    const double t0b = fma(h[1], yb, h[0] * xb) + h[2];
    const double t1b = fma(h[4], yb, h[3] * xb) + h[5];
    const double t2b = fma(h[7], yb, h[6] * xb) + h[8];
    // sums of magnitudes over the cell: what the float64 roundings of the reference scale with
    const double S0 = fabs(h[0]) * (fabs(xb) + DX) + fabs(h[1]) * (fabs(yb) + DY) + fabs(h[2]);
    const double S1 = fabs(h[3]) * (fabs(xb) + DX) + fabs(h[4]) * (fabs(yb) + DY) + fabs(h[5]);
    const double S2 = fabs(h[6]) * (fabs(xb) + DX) + fabs(h[7]) * (fabs(yb) + DY) + fabs(h[8]);
    const double at2 = fabs(t2b);
    const double g = fabs(h[6]) * DX + fabs(h[7]) * DY;      // how far t2 moves inside the cell
    bool ok = origin_ok && at2 > 1e-20 && at2 < 1e20 && g <= 0.25 * at2;
    const double tmin = at2 - g, rho = (at2 + g) / tmin;
    const double Qx = t0b / t2b, Qy = t1b / t2b;
    ok = ok && fabs(Qx) < 1073741824.0 && fabs(Qy) < 1073741824.0;
    const double n0x = floor(Qx), n0y = floor(Qy);
    const double Ax = fma(-Qx, h[6], h[0]), Bx = fma(-Qx, h[7], h[1]);
    const double Ay = fma(-Qy, h[6], h[3]), By = fma(-Qy, h[7], h[4]);
    // |s| / 2^16 <= (|c'| + |a'| DX + |b'| DY) / tmin <= M + f (at2 + g) / tmin,  f < 1.25
    const double Mx = (fabs(Ax) * DX + fabs(Bx) * DY) / tmin, My = (fabs(Ay) * DX + fabs(By) * DY) / tmin;
    const double Smax = fmax(Mx, My) + 1.25 * rho;
    // The estimate is a 16.16 fixed-point number: its integer half is a SIGNED 16 bits, so |estimate| must stay far below
    // 2^15 pixels - a saturated v_cvt_flr_i32_f32 (INT_MAX: fraction 0xffff) would pass the doubt test.  kFastMaxEstimate
    // keeps it two powers of two inside; cells whose bound is larger take the exact path for every pixel.
    constexpr double kFastMaxEstimate = 500.0;
    static_assert(kFastMaxEstimate * 4 < (double)(1 << (31 - kFastFracBits)), "the estimate's integer half must fit the fixed-point format");
    ok = ok && Smax < kFastMaxEstimate;
    // float32 side, relative to |s|: 3 (inputs + two FMA roundings of the numerator) + 3 rho (the same of
    // the denominator) + 2 (v_rcp_f32, 1 ulp) + 1 (the product); 0.5 on top for second-order terms
    const double E32 = (6.5 + 3.0 * rho) * eps32 * Smax;
    // float64 side: the reference's own roundings of t0, t2 and of the quotient, the residual R of Q,
    // the roundings of A, B - all <= 16 eps64 (S + (|Q| + Smax)(S2 + |t2b|)) / tmin
    const double E64 = 16.0 * eps64 * (fmax(S0, S1) + (fmax(fabs(Qx), fabs(Qy)) + Smax) * (S2 + at2)) / tmin;
    // the shift: strictly above the bound; the window in units of 2^-16, at least one
    const double dE = (E32 + E64) * (1.0 + 9.5367431640625e-07) + 9.094947017729282e-13;     // (1 + 2^-20), 2^-40
    const double t16 = fmax(ceil(2.0 * dE * unit), 1.0);
    ok = ok && dE < 0.125;               // (NaN fails)
    if (ok) {
        const double fx = (Qx - n0x) + dE, fy = (Qy - n0y) + dE;
        pa.x = (float)(unit * fma(fx, h[6], Ax));
        pa.y = (float)(unit * fma(fx, h[7], Bx));
        pa.z = (float)(unit * (fx * t2b));
        pa.w = (float)(unit * fma(fy, h[6], Ay));
        pb.x = (float)(unit * fma(fy, h[7], By));
        pb.y = (float)(unit * (fy * t2b));
        pb.z = (float)t2b;
        pb.w = (float)h[6];
        pc.x = (float)h[7];
        pc.y = __int_as_float((int)n0x);
        pc.z = __int_as_float((int)n0y);
        pc.w = __uint_as_float((unsigned)t16);
    }
    out[0] = pa;
    out[1] = pb;
    out[2] = pc;
    if (src_rows) {
        const int m = ok ? (int)ceil(Smax) + 2 : 0;
        atomicMin(src_rows, ok ? (int)n0y - m : -2147483647 - 1);
        atomicMax(src_rows + 1, ok ? (int)n0y + m : 2147483647);
    }
}

// Workspace of the warp (every part rounded up to 256 bytes): first what depends on the mesh edges, the canvas and
// the offsets only and is SHARED by all pairs of a batch,
//   lut [final_h + final_w] i32 + 1 stamp word | fcol [final_w rounded up to 4] u32 | frow [final_h] uint2 | src_rows [2 rows + 2] i32
// then, per pair of the batch, what depends on that pair's H grid,
//   hinv_pad [cells][10] f64 | frec [(rows + 1)(cols + 1)][3] float4
// frec, fcol, frow serve k_warp_fast: record of cell (r, c) at r (cols + 1) + c, row `rows` and column
// `cols` hold the everything-in-doubt record; fcol[j] = cell column | (dx + 128) << 16, frow[i] = {cell row | stamp16 << 16,
// float bits of dy}, dx / dy = signed distance from the cell's anchor (its middle pixel); pixels that are not
// inside an ordinary cell point at the extra row / column.
struct WarpWork {
    double *hinv_pad;   // pair 0; pair k at + k * hinv_stride doubles
    int *lut;
    float4 *frec;       // pair 0; pair k at + k * frec_stride float4
    unsigned *fcol;
    uint2 *frow;
    int *src_rows;      // [rows][2] source-row interval per cell row, then 1 flag word (bit 0: irregular mesh)
    long long hinv_stride, frec_stride;
    size_t bytes;
};

// distance between consecutive pairs of a batch: bytes for the images, doubles / float4 for the per-pair tables
struct WarpStrides {
    long long img, out, center, hinv, frec;
};

__host__ __device__ inline size_t round256(size_t b) { return (b + 255) / 256 * 256; }

inline WarpWork warp_work_layout(void *base, int mesh_rows, int mesh_cols, int final_w, int final_h, int batch = 1) {
    WarpWork w;
    char *p = (char *)base;
    const size_t cells = (size_t)mesh_rows * mesh_cols;
    w.lut = (int *)p;
    p += round256(((size_t)final_w + final_h + 1) * sizeof(int));      // + the stamp word (warp_stamp)
    w.fcol = (unsigned *)p;
    p += round256((((size_t)final_w + 3) / 4 * 4) * sizeof(unsigned));
    w.frow = (uint2 *)p;
    p += round256((size_t)final_h * sizeof(uint2));
    w.src_rows = (int *)p;
    p += round256(((size_t)mesh_rows * 2 + 2) * sizeof(int));
    const size_t hinv_bytes = round256(cells * APAP_HINV_STRIDE * sizeof(double));
    const size_t frec_bytes = round256(((size_t)mesh_rows + 1) * ((size_t)mesh_cols + 1) * 3 * sizeof(float4));
    w.hinv_pad = (double *)p;
    w.frec = (float4 *)(p + hinv_bytes);
    w.hinv_stride = (long long)((hinv_bytes + frec_bytes) / sizeof(double));
    w.frec_stride = (long long)((hinv_bytes + frec_bytes) / sizeof(float4));
    p += (hinv_bytes + frec_bytes) * (size_t)(batch < 1 ? 1 : batch);
    w.bytes = (size_t)(p - (char *)base);
    return w;
}

// entry `e` of the extra row / column of the record tables: everything in doubt, no exact floats
__device__ __forceinline__ void warp_extra_entry(float4 *__restrict__ frec, size_t e) {
    const double z[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    fast_record(z, false, 0.0, 0.0, 1.0, 1.0, frec + e * 3);
}

// The warp's per-cell tables of mesh cell (er, ec) from its FORWARD matrix `m` (the grid's values, widened): the inverse in
// the grid's dtype T (apap.py:201-203), padded for the exact path, its float32-estimate record, its exact-path floats.  One
// function for k_warp_setup (the stored grid) and for the solve's tail (the value it has just rounded to float32).
template <typename T>
__device__ __forceinline__ void warp_cell_tables(const double (&m)[9], int er, int ec, int mesh_cols, const double *__restrict__ mesh_w,
                                                 int n_w, const double *__restrict__ mesh_h, int n_h, int final_w, int final_h,
                                                 int off_x, int off_y, double *__restrict__ hinv_pad, T *__restrict__ hinv_dense,
                                                 float4 *__restrict__ frec, int *status, int *src_rows,
                                                 const CellEdges *edges = nullptr) {
    const int cell = er * mesh_cols + ec;
    const size_t e = (size_t)er * (mesh_cols + 1) + ec;
    double r[9];
    int x0, y0, sx, sy;      // the edge loads travel with the matrix loads (or were fetched earlier: the solve's tail)
    const bool okx = edges ? fast_origin_of(edges->w0, edges->w1, ec >= 0 && ec + 1 < n_w, final_w, x0, sx)
                           : fast_origin(mesh_w, n_w, ec, final_w, x0, sx);
    const bool oky = edges ? fast_origin_of(edges->h0, edges->h1, er >= 0 && er + 1 < n_h, final_h, y0, sy)
                           : fast_origin(mesh_h, n_h, er, final_h, y0, sy);
    if (!inv3(m, r)) atomicOr(status, apap::kStatusSingular);
    double2 *p = reinterpret_cast<double2 *>(hinv_pad + (size_t)cell * APAP_HINV_STRIDE);
    // the inverse rounded to the grid's dtype (what the reference stores back, apap.py:203),
    // widened once here instead of nine v_cvt_f64_f32 per pixel in the warp kernel
    double hd[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) hd[k] = (double)(T)r[k];
    p[0] = make_double2(hd[0], hd[1]);
    p[1] = make_double2(hd[2], hd[3]);
    p[2] = make_double2(hd[4], hd[5]);
    p[3] = make_double2(hd[6], hd[7]);
    p[4] = make_double2(hd[8], 0.0);
    if (hinv_dense) {
#pragma unroll
        for (int k = 0; k < 9; ++k) hinv_dense[(size_t)cell * 9 + k] = (T)r[k];
    }
    // anchor in the middle of the cell: the estimate's error grows with the distance from it
    fast_record(hd, okx && oky, (double)(x0 + sx / 2 - off_x), (double)(y0 + sy / 2 - off_y), (double)(sx - sx / 2),
                (double)(sy - sy / 2), frec + e * 3, src_rows ? src_rows + 2 * er : nullptr);
}

// the solve's tail (eigen_denorm_cell) leaves its cell warp ready; the last column / row also write the extra entries
__device__ void warp_emit_from_solve(const WarpEmit &we, int pair, int cell, const float (&Hf)[9], const CellEdges &edges) {
    double m[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) m[k] = (double)Hf[k];
    const int er = cell / we.mesh_cols, ec = cell - er * we.mesh_cols;
    double *hinv_pad = we.hinv_pad + (long long)pair * we.hinv_stride;
    float4 *frec = we.frec + (long long)pair * we.frec_stride;
    warp_cell_tables<float>(m, er, ec, we.mesh_cols, we.mesh_w, we.n_w, we.mesh_h, we.n_h, we.final_w, we.final_h, we.off_x, we.off_y,
                            hinv_pad, (float *)nullptr, frec, we.status, (int *)nullptr, &edges);
    const size_t stride = (size_t)we.mesh_cols + 1;
    if (ec == we.mesh_cols - 1) warp_extra_entry(frec, (size_t)er * stride + we.mesh_cols);
    if (er == we.mesh_rows - 1) {
        warp_extra_entry(frec, (size_t)we.mesh_rows * stride + ec);
        if (ec == we.mesh_cols - 1) warp_extra_entry(frec, (size_t)we.mesh_rows * stride + we.mesh_cols);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_warp_setup(const T *__restrict__ H, int cells,
                                                    double *__restrict__ hinv_pad,
                                                    T *__restrict__ hinv_dense, int inv_blocks,
                                                    const double *__restrict__ mesh_w, int n_w,
                                                    const double *__restrict__ mesh_h, int n_h,
                                                    int mesh_rows, int mesh_cols, int final_w, int final_h,
                                                    int *__restrict__ lut, int *status, int off_x, int off_y,
                                                    float4 *__restrict__ frec, unsigned *__restrict__ fcol,
                                                    uint2 *__restrict__ frow, int *__restrict__ src_rows,
                                                    long long hinv_stride, long long frec_stride) {
    // grid.y = pair of a batch (every pair has the same mesh and canvas: the lookup tables are built once, by
    // y = 0); blocks [0, inv_blocks) are the per-cell half - what depends on the H grid -, the rest the tables.
    // Either half may be absent from a launch (inv_blocks = 0, or a grid of inv_blocks blocks).
    // src_rows (optional; the host-buffer warp schedules its upload / warp / download bands with it): per
    // cell row the interval of source rows its pixels can read, [2 r] pre-set to INT_MAX-like, [2 r + 1] to
    // INT_MIN-like by the caller; word [2 rows]: bit 0 set when some canvas pixel is not inside an ordinary cell
    // (then the intervals say nothing).
    __shared__ double pm[2][kMaxEdges];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < inv_blocks) {
        // one thread per entry of the (rows + 1) x (cols + 1) record table; the extra row and column are the
        // everything-in-doubt records.  (Inverting each cell twice, in two waves - one storing the inverse, one
        // building the record - to shorten the dependent chain measured slower: 8.6 vs 7.6 us.)
        const size_t pair = blockIdx.y;
        H += pair * (size_t)cells * 9;
        hinv_pad += pair * hinv_stride;
        frec += pair * frec_stride;
        if (hinv_dense) hinv_dense += pair * (size_t)cells * 9;
        const int e = blockIdx.x * 256 + tid;
        const int er = e / (mesh_cols + 1), ec = e - er * (mesh_cols + 1);
        if (er > mesh_rows) return;
        if (er == mesh_rows || ec == mesh_cols) {
            warp_extra_entry(frec, e);
            return;
        }
        const int cell = er * mesh_cols + ec;
        double m[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k] = (double)H[(size_t)cell * 9 + k];
        warp_cell_tables<T>(m, er, ec, mesh_cols, mesh_w, n_w, mesh_h, n_h, final_w, final_h, off_x, off_y, hinv_pad, hinv_dense,
                            frec, status, src_rows);
        return;
    }
    if (blockIdx.y != 0) return;
    const int row_blocks = (final_h + 1023) / 1024;
    const int b = blockIdx.x - inv_blocks;
    if (b == 0 && tid == 0) lut[final_w + final_h] = warp_stamp(mesh_rows, mesh_cols, final_w, final_h);
    const bool is_row = b < row_blocks;
    const double *edges = is_row ? mesh_h : mesh_w;
    const int n_e = is_row ? n_h : n_w;
    const int ncell = is_row ? mesh_rows : mesh_cols;
    const int count = is_row ? final_h : final_w;
    const int base = (is_row ? b : b - row_blocks) * 1024;
    // inclusive max-scan (Hillis-Steele, ping-pong buffers)
    for (int i = tid; i < n_e; i += 256) pm[0][i] = edges[i];
    __syncthreads();
    int cur = 0;
    for (int off = 1; off < n_e; off <<= 1) {
        for (int i = tid; i < n_e; i += 256) {
            const double v = pm[cur][i];
            pm[cur ^ 1][i] = (i >= off) ? fmax(v, pm[cur][i - off]) : v;
        }
        __syncthreads();
        cur ^= 1;
    }
    const double *run = pm[cur];
    for (int q = tid; q < 1024; q += 256) {
        const int idx = base + q;
        if (idx >= count) break;
        const double x = (double)idx;
        int lo = 0, hi = n_e;  // first k in [0, n_e) with x < run[k]; n_e = none
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (x < run[mid]) hi = mid; else lo = mid + 1;
        }
        int c = lo - 1;
        bool bad = lo >= n_e;
        if (c < 0) c += ncell;  // Python's index -1
        bad = bad || c < 0 || c >= ncell;
        if (bad) { atomicOr(status, apap::kStatusIndex); c = 0; }
        lut[(is_row ? 0 : final_h) + idx] = c;
        // the fast kernel's entry: cell and distance from the cell's anchor, or the extra cell
        int x0 = 0, span = 1;
        const bool ok = !bad && fast_origin(edges, n_e, c, count, x0, span);
        const int d = idx - x0;
        const bool in = ok && d >= 0 && d < span;
        const unsigned fc = in ? (unsigned)c : (unsigned)ncell;
        if (!in && src_rows) atomicOr(src_rows + 2 * mesh_rows, 1);
        const int rel = in ? d - span / 2 : 0;       // distance from the cell's anchor (its middle pixel), -127 ... 127
        const unsigned fd = (unsigned)(rel + 128);
        if (is_row) {
            frow[idx] = make_uint2(fc | (warp_stamp16(mesh_rows, mesh_cols, final_w, final_h) << 16), __float_as_uint((float)rel));
        } else {
            fcol[idx] = fc | (fd << 16);
            // the pad up to a multiple of 4 columns repeats the last column (pixels past the row end are
            // computed like the last one and not stored)
            if (idx == count - 1)
                for (int q = idx + 1; q < (count + 3) / 4 * 4; ++q) fcol[q] = fc | (fd << 16);
        }
    }
}

// 12 bytes to ANY byte address with the non-temporal hint (global_store_dwordx3 ... nt)
__device__ __forceinline__ void store12_stream(uint8_t *p, unsigned a, unsigned b, unsigned c) {
    typedef unsigned Dwords3 __attribute__((ext_vector_type(3)));
    typedef Dwords3 Dwords3AnyByte __attribute__((aligned(1)));
    const Dwords3 v = {a, b, c};
    __builtin_nontemporal_store(v, reinterpret_cast<Dwords3AnyByte *>(p));
}

// target coordinate of canvas pixel (i, j) through the (already inverted) cell matrix:
// float64 FMA chain in the order h0*x + h1*y + h2, then the two divisions by the third
// component (apap.py:172-184,211-213: float32 H^-1 promoted to float64 by the int64
// point).  The two quotients share one reciprocal (v_rcp_f64 + two Newton steps) and get
// one residual correction each - the Markstein sequence, which returns the correctly
// rounded quotient the reference's true division produces (checked against the oracle's
// coordinates in tests: equal).
struct Hinv9 {
    double2 a, b, c, d, e;  // h0 h1 | h2 h3 | h4 h5 | h6 h7 | h8 -
};

__device__ __forceinline__ Hinv9 load_hinv(const double *__restrict__ hinv_pad, unsigned cell) {
    // unsigned 32-bit byte offset: lets the load use the scalar-base + 32-bit-offset form
    const double2 *p = reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(hinv_pad) +
                                                         (size_t)(cell * (unsigned)(APAP_HINV_STRIDE * sizeof(double))));
    Hinv9 h;
    h.a = p[0]; h.b = p[1]; h.c = p[2]; h.d = p[3]; h.e = p[4];
    return h;
}

__device__ __forceinline__ Hinv9 select_hinv(bool first, const Hinv9 &x, const Hinv9 &y) {
    Hinv9 h;
    h.a = first ? x.a : y.a; h.b = first ? x.b : y.b; h.c = first ? x.c : y.c;
    h.d = first ? x.d : y.d; h.e.x = first ? x.e.x : y.e.x; h.e.y = 0.0;
    return h;
}

__device__ __forceinline__ void target_from(const Hinv9 &h, double x, double y, double &tx, double &ty) {
    const double t0 = fma(h.b.x, 1.0, fma(h.a.y, y, h.a.x * x));
    const double t1 = fma(h.c.y, 1.0, fma(h.c.x, y, h.b.y * x));
    const double t2 = fma(h.e.x, 1.0, fma(h.d.y, y, h.d.x * x));
    double r = __builtin_amdgcn_rcp(t2);
    r = fma(fma(-t2, r, 1.0), r, r);
    r = fma(fma(-t2, r, 1.0), r, r);
    const double q0 = t0 * r, q1 = t1 * r;
    tx = fma(fma(-t2, q0, t0), r, q0);
    ty = fma(fma(-t2, q1, t1), r, q1);
}

__device__ __forceinline__ void target_of(const double *__restrict__ hinv_pad, int cell, double x,
                                          double y, double &tx, double &ty) {
    target_from(load_hinv(hinv_pad, (unsigned)cell), x, y, tx, ty);
}

// K3, flat-order form (fallback for sources with a side of 2^24 pixels or more, and the
// baseline of the strip form below).  One thread = 4 consecutive canvas pixels in flat order = 12
// contiguous output bytes = one global_store_dwordx3; a wave writes 768 contiguous
// bytes.  The gather reads 3 bytes per pixel with one unaligned dword load; neighbouring
// lanes read neighbouring source pixels because local homographies are close to the
// global one.  The body is branch-free and staged (all lookups, then all H^-1 loads,
// then arithmetic, then all gathers) so that a thread keeps 8 / 12 / 4 loads in flight
// instead of walking a dependent chain per pixel: the kernel is latency-bound, not
// issue-bound.  Requires img_bytes >= 4.
// kBlend fuses the stitch the reference's __main__ has commented out (apap.py:258-262):
// paste the centre image at (off_x, off_y), then uniform_blend (apap_utils.py:75-88) - in
// the same pass, so the warped canvas is never written and re-read.
template <bool kBlend>
__global__ __launch_bounds__(256) void k_warp(const uint8_t *__restrict__ img, int img_h, int img_w,
                                              const double *__restrict__ hinv_pad, int mesh_cols,
                                              const int *__restrict__ lut, int final_w, int final_h,
                                              int off_x, int off_y, uint8_t *__restrict__ out,
                                              const uint8_t *__restrict__ center, int center_h, int center_w,
                                              int row_begin, int row_count, WarpStrides st, int mesh_rows, int *status) {
    if (!warp_tables_ready(lut[final_w + final_h], mesh_rows, mesh_cols, final_w, final_h, status)) return;
    // The launch covers canvas rows [row_begin, row_begin + row_count) - the whole canvas,
    // or one rank's band when a pair is sharded over GPUs; `out` points at the band's first row.
    // grid.z = pair of a batch (one mesh and canvas geometry, its own image, grid and canvas).
    {
        const long long pair = blockIdx.z;
        img += pair * st.img;
        out += pair * st.out;
        if (kBlend) center += pair * st.center;
        hinv_pad += pair * st.hinv;
    }
    // 32-bit pixel indices (the launcher refuses canvases of 2^31 pixels or more): a
    // 64-bit division here expands into ~100 instructions with branches
    const unsigned total = (unsigned)final_w * (unsigned)row_count;
    const unsigned last = (unsigned)img_h * (unsigned)img_w * 3u - 4u;
    const unsigned g = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (g >= total) return;
    int ii[4], jj[4];
    {
        int i = (int)(g / (unsigned)final_w);
        int j = (int)(g - (unsigned)i * (unsigned)final_w);
        i += row_begin;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ii[k] = min(i, row_begin + row_count - 1);  // only the last group can run past the band
            jj[k] = j;
            if (++j == final_w) { j = 0; ++i; }
        }
    }
    // Lookups.  The four pixels share a canvas row unless the group wraps, so two row
    // lookups (first and last pixel) serve all four.
    const int r0 = lut[(unsigned)ii[0]], r3 = lut[(unsigned)ii[3]];
    int cell[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        cell[k] = (ii[k] == ii[0] ? r0 : r3) * mesh_cols + lut[(unsigned)(final_h + jj[k])];
    // H^-1 of the first and the last pixel's cells; the two in between almost always sit in
    // one of those (cell indices are monotone along a row and cells are wider than 2 px).
    Hinv9 hv[4];
    hv[0] = load_hinv(hinv_pad, (unsigned)cell[0]);
    hv[3] = load_hinv(hinv_pad, (unsigned)cell[3]);
#pragma unroll
    for (int k = 1; k < 3; ++k) {
        const bool is_a = cell[k] == cell[0];
        hv[k] = select_hinv(is_a, hv[0], hv[3]);
        if (!is_a && cell[k] != cell[3]) hv[k] = load_hinv(hinv_pad, (unsigned)cell[k]);  // a third cell inside four pixels
    }
    unsigned off[4];  // byte offsets into the image: the launcher refuses images of 4 GiB or more
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double tx, ty;
        target_from(hv[k], (double)(jj[k] - off_x), (double)(ii[k] - off_y), tx, ty);
        // strict inequalities and truncation, apap.py:214-215; NaN fails every comparison
        ok[k] = 0.0 < tx && tx < (double)img_w && 0.0 < ty && ty < (double)img_h;
        const int ix = ok[k] ? (int)tx : 0, iy = ok[k] ? (int)ty : 0;
        off[k] = ((unsigned)iy * (unsigned)img_w + (unsigned)ix) * 3u;
    }
    unsigned int px[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // the dword at the pixel's first byte; for the image's very last pixel read the
        // dword one byte earlier and shift, so that no byte beyond the image is touched
        unsigned int v;
        const unsigned oc = off[k] < last ? off[k] : last;
        __builtin_memcpy(&v, img + oc, 4);
        v >>= 8 * (off[k] - oc);
        px[k] = ok[k] ? (v & 0x00ffffffu) : 0u;
    }
    if (kBlend) {
        const unsigned clast = (unsigned)center_h * (unsigned)center_w * 3u - 4u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ci = ii[k] - off_y, cj = jj[k] - off_x;
            const bool in = ci >= 0 && ci < center_h && cj >= 0 && cj < center_w;
            const unsigned co = in ? ((unsigned)ci * (unsigned)center_w + (unsigned)cj) * 3u : 0u;
            const unsigned cc = co < clast ? co : clast;
            unsigned int c;
            __builtin_memcpy(&c, center + cc, 4);
            c = in ? ((c >> (8 * (co - cc))) & 0x00ffffffu) : 0u;
            // uniform_blend: a pixel is "present" when its channel mean is > 0, i.e. any
            // channel is non-zero; both present -> floor((a + b) / 2) per channel (float64
            // sum * 0.5, astype(uint8)), otherwise a + b with one of them 0
            const unsigned w = px[k];
            const unsigned avg = (w & c) + (((w ^ c) & 0x00fefefeu) >> 1);
            px[k] = (w != 0u && c != 0u) ? avg : (w | c);
        }
    }
    uint8_t *o = out + (size_t)g * 3;
    if (g + 4u <= total) {
        store12_stream(o, px[0] | (px[1] << 24), (px[1] >> 8) | (px[2] << 16), (px[2] >> 16) | (px[3] << 8));
    } else {
        for (unsigned k = 0; k < 4u && g + k < total; ++k) {
            o[3 * k] = (uint8_t)(px[k] & 0xff);
            o[3 * k + 1] = (uint8_t)((px[k] >> 8) & 0xff);
            o[3 * k + 2] = (uint8_t)((px[k] >> 16) & 0xff);
        }
    }
}

// K3, row-strip form.  Mesh cells are small (C3: ~20 x 11 pixels), so the flat-order kernel
// above fetches two 72-byte matrices for every 12 bytes it writes.  Here a lane owns 4
// consecutive pixels of a row and walks kRows consecutive rows with them; the wave shares the
// row, so the row's cell lookup is a scalar load, the lane's four column lookups are done once,
// and the matrices are re-fetched only when the strip crosses into the next cell row (a
// wave-uniform branch).  Same arithmetic per pixel, same bytes written.
// What a lane keeps per pixel while its strip stays in one cell row: the three products with
// the pixel's x (the first terms of the reference's sums, apap.py:172-184) and the other six
// coefficients.
struct PixelH {
    double p0, p1, p2;  // h0 x, h3 x, h6 x
    double h1, h2, h4, h5, h7, h8;
};

__device__ __forceinline__ PixelH pixel_h(const Hinv9 &h, double x) {
    PixelH q;
    q.p0 = h.a.x * x; q.p1 = h.b.y * x; q.p2 = h.d.x * x;
    q.h1 = h.a.y; q.h2 = h.b.x; q.h4 = h.c.x; q.h5 = h.c.y; q.h7 = h.d.y; q.h8 = h.e.x;
    return q;
}

template <bool kBlend, int kRows>
__global__ __launch_bounds__(256) void k_warp_rows(const uint8_t *__restrict__ img, int img_h, int img_w,
                                                   const double *__restrict__ hinv_pad, int mesh_cols,
                                                   const int *__restrict__ lut, int final_w, int final_h,
                                                   int off_x, int off_y, uint8_t *__restrict__ out,
                                                   const uint8_t *__restrict__ center, int center_h, int center_w,
                                                   int row_begin, int row_count, WarpStrides st, int mesh_rows, int *status) {
    if (!warp_tables_ready(lut[final_w + final_h], mesh_rows, mesh_cols, final_w, final_h, status)) return;
    // (Renumbering the blocks so that each XCD owns a band of rows was measured: no change.)
    {
        const long long pair = blockIdx.z;      // pair of a batch
        img += pair * st.img;
        out += pair * st.out;
        if (kBlend) center += pair * st.center;
        hinv_pad += pair * st.hinv;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j0 = ((int)blockIdx.x * 64 + lane) * 4;
    const int y_first = row_begin + ((int)blockIdx.y * 4 + wave) * kRows;
    const int y_end = min(y_first + kRows, row_begin + row_count);
    if (j0 >= final_w || y_first >= y_end) return;
    const unsigned last = (unsigned)img_h * (unsigned)img_w * 3u - 4u;
    const unsigned clast = kBlend ? (unsigned)center_h * (unsigned)center_w * 3u - 4u : 0u;
    const int npx = min(4, final_w - j0);
    int col[4];
    double xs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = min(j0 + k, final_w - 1);  // pixels past the row end repeat the last one and are not stored
        col[k] = lut[(unsigned)(final_h + j)];
        xs[k] = (double)(j - off_x);
    }
    // Staged over the whole strip so that a lane has its kRows x 4 gathers in flight at once
    // instead of a lookup -> matrices -> arithmetic -> gather -> store chain per row.
    int rr[kRows];
#pragma unroll
    for (int t = 0; t < kRows; ++t) rr[t] = lut[(unsigned)min(y_first + t, y_end - 1)];
#pragma unroll
    for (int t = 0; t < kRows; ++t) rr[t] = __builtin_amdgcn_readfirstlane(rr[t]);
    unsigned off[kRows][4];
    // One pass per cell row the strip touches (usually one, two when it crosses an edge): fetch
    // that row's matrices, then do every strip row that lies in it.  All branches are wave-uniform.
    unsigned todo = (1u << kRows) - 1u;
    while (todo != 0u) {
        const int first = __builtin_ctz(todo);
        int r = rr[0];
#pragma unroll
        for (int t = 1; t < kRows; ++t) r = (t == first) ? rr[t] : r;
        const int base = r * mesh_cols;
        const Hinv9 ha = load_hinv(hinv_pad, (unsigned)(base + col[0]));
        const Hinv9 hb = load_hinv(hinv_pad, (unsigned)(base + col[3]));
        PixelH q[4];
        q[0] = pixel_h(ha, xs[0]);
        q[3] = pixel_h(hb, xs[3]);
#pragma unroll
        for (int k = 1; k < 3; ++k) {
            const bool is_a = col[k] == col[0];
            Hinv9 hk = select_hinv(is_a, ha, hb);
            if (!is_a && col[k] != col[3]) hk = load_hinv(hinv_pad, (unsigned)(base + col[k]));  // a third cell
            q[k] = pixel_h(hk, xs[k]);
        }
#pragma unroll
        for (int t = 0; t < kRows; ++t) {
            if (rr[t] != r) continue;
            todo &= ~(1u << t);
            const double yd = (double)(y_first + t - off_y);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // (h0 x + h1 y) + h2 and so on: the order of the reference's matrix-vector product
                const double t0 = fma(q[k].h1, yd, q[k].p0) + q[k].h2;
                const double t1 = fma(q[k].h4, yd, q[k].p1) + q[k].h5;
                const double t2 = fma(q[k].h7, yd, q[k].p2) + q[k].h8;
                // shared reciprocal (one Newton step: ~2^-46) and a residual correction per
                // quotient: the quotient's error before its final rounding is ~2^-92
                double rc = __builtin_amdgcn_rcp(t2);
                rc = fma(fma(-t2, rc, 1.0), rc, rc);
                const double q0 = t0 * rc, q1 = t1 * rc;
                const double tx = fma(fma(-t2, q0, t0), rc, q0);
                const double ty = fma(fma(-t2, q1, t1), rc, q1);
                // strict 0 < t < size, then truncation (apap.py:214-215).  For t > 0 the upper
                // test is the same on the truncated integer (the conversion saturates, NaN
                // fails t > 0).
                const int ix = (int)tx, iy = (int)ty;
                const bool ok = (tx > 0.0) & (ty > 0.0) & (ix < img_w) & (iy < img_h);  // no short-circuit branches
                // a pixel outside the source is marked by the sign bit (the launcher sends
                // sources of 2 GiB or more to the flat-order kernel)
                off[t][k] = ok ? (__umul24((unsigned)iy, (unsigned)img_w) + (unsigned)ix) * 3u : 0xffffffffu;
            }
        }
    }
    unsigned int px[kRows][4];
#pragma unroll
    for (int t = 0; t < kRows; ++t) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // the dword at the pixel's first byte; for the image's very last pixel the dword one
            // byte earlier, shifted (v_alignbyte_b32), so that no byte beyond the image is touched
            unsigned int v;
            const unsigned o = off[t][k];
            const unsigned oc = o < last ? o : last;
            __builtin_memcpy(&v, img + oc, 4);
            v = __builtin_amdgcn_alignbyte(0u, v, o - oc);
            // v & 0xffffff & ~sign(o): v_bfe_i32 + v_bitop3_b32 (truth table a & b & ~c = 0x40)
            px[t][k] = (unsigned)__builtin_amdgcn_bitop3_b32((int)v, 0x00ffffff, __builtin_amdgcn_sbfe((int)o, 31u, 1u), 0x40);
        }
    }
#pragma unroll
    for (int t = 0; t < kRows; ++t) {
        const int y = y_first + t;
        if (y >= y_end) break;  // wave-uniform
        if (kBlend) {
            const int ci = y - off_y;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cj = j0 + k - off_x;
                const bool in = ci >= 0 && ci < center_h && cj >= 0 && cj < center_w;
                const unsigned co = in ? ((unsigned)ci * (unsigned)center_w + (unsigned)cj) * 3u : 0u;
                const unsigned cc = co < clast ? co : clast;
                unsigned int c;
                __builtin_memcpy(&c, center + cc, 4);
                c = in ? ((c >> (8 * (co - cc))) & 0x00ffffffu) : 0u;
                const unsigned w = px[t][k];
                const unsigned avg = (w & c) + (((w ^ c) & 0x00fefefeu) >> 1);
                px[t][k] = (w != 0u && c != 0u) ? avg : (w | c);
            }
        }
        uint8_t *o = out + ((size_t)(y - row_begin) * (size_t)final_w) * 3 + (unsigned)j0 * 3u;
        if (npx == 4) {
            // 4 x 24 bits -> 3 dwords: one shift-or and two byte permutes (v_perm_b32 picks bytes 0-3 from its second
            // operand, 4-7 from its first); rows start at any byte: an unaligned, non-temporal 12-byte store
            store12_stream(o, px[t][0] | (px[t][1] << 24), __builtin_amdgcn_perm(px[t][2], px[t][1], 0x05040201u),
                           __builtin_amdgcn_perm(px[t][3], px[t][2], 0x06050402u));
        } else {
            for (int k = 0; k < npx; ++k) {
                o[3 * k] = (uint8_t)(px[t][k] & 0xff);
                o[3 * k + 1] = (uint8_t)((px[t][k] >> 8) & 0xff);
                o[3 * k + 2] = (uint8_t)((px[t][k] >> 16) & 0xff);
            }
        }
    }
}

// K3, default form: the row strips of k_warp_rows with the float32 estimate described at fast_record.
// Same unit of work (a lane owns 4 consecutive pixels of a canvas row and walks kRows rows with them),
// same gathers and stores; what differs is how a pixel's source offset is found:
//   * the lane's four column entries arrive in ONE 16-byte load (cell column | dx << 16), the strip's
//     row entries by scalar loads ({cell row, dy as float}: the rows are wave-uniform);
//   * per cell row the strip touches, two 48-byte records (first and last pixel's cells) instead of two
//     80-byte matrices; per pixel and row 3 float32 FMAs, one v_rcp_f32, two products, two conversions
//     and integer work - no float64 instruction at all;
//   * a pixel whose fixed-point estimate has its low 22 bits inside the cell's doubt window (a few in
//     10^5 on BASELINE's configurations) is recomputed with the exact float64 sequence - the same code
//     k_warp_rows runs for every pixel - in a loop that only waves holding such a pixel enter.
// 40 instead of 72 registers of per-pixel constants, 6-8 waves per SIMD instead of 4.
// the 3 bytes at byte offset `o` of the source as a 24-bit value; 0 for the "outside" marker 0xffffffff.
// Reads the dword at the pixel's first byte; for the image's very last pixel the dword one byte earlier,
// shifted (v_alignbyte_b32), so that no byte beyond the image is touched.
__device__ __forceinline__ unsigned gather_px(const uint8_t *__restrict__ img, unsigned o, unsigned last) {
    unsigned int v;
    const unsigned oc = o < last ? o : last;
    __builtin_memcpy(&v, img + oc, 4);
    v = __builtin_amdgcn_alignbyte(0u, v, o - oc);
    // v & 0xffffff & ~sign(o): v_bfe_i32 + v_bitop3_b32 (truth table a & b & ~c = 0x40)
    return (unsigned)__builtin_amdgcn_bitop3_b32((int)v, 0x00ffffff, __builtin_amdgcn_sbfe((int)o, 31u, 1u), 0x40);
}

// floor(v) as an int32 in ONE instruction (saturating, NaN -> 0): hipcc emits v_floor_f32 + v_cvt_i32_f32 for (int)floorf(v)
__device__ __forceinline__ int floor_to_int(float v) {
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

// `cr`, `cc`: the pixel's cell row and column as the fast tables give them - the exact table's values unless they
// point at the extra row / column (then the exact table is read: one more memory round trip, irregular meshes only)
__device__ __forceinline__ unsigned exact_offset(const double *__restrict__ hinv_pad, const int *__restrict__ lut, int mesh_rows,
                                                 int mesh_cols, int final_h, unsigned cr, unsigned cc, int i, int j, int off_x,
                                                 int off_y, int img_w, int img_h) {
    if (cr >= (unsigned)mesh_rows) cr = (unsigned)lut[(unsigned)i];
    if (cc >= (unsigned)mesh_cols) cc = (unsigned)lut[(unsigned)(final_h + j)];
    const int cell = (int)(cr * (unsigned)mesh_cols + cc);
    double tx, ty;
    target_of(hinv_pad, cell, (double)(j - off_x), (double)(i - off_y), tx, ty);
    const int ix = (int)tx, iy = (int)ty;
    const bool ok = (tx > 0.0) & (ty > 0.0) & (ix < img_w) & (iy < img_h);   // as in k_warp_rows
    return ok ? (__umul24((unsigned)iy, (unsigned)img_w) + (unsigned)ix) * 3u : 0xffffffffu;
}

template <bool kBlend, int kRows>
__global__ __launch_bounds__(256) void k_warp_fast(const uint8_t *__restrict__ img, int img_h, int img_w,
                                                   const double *__restrict__ hinv_pad, int mesh_rows, int mesh_cols,
                                                   const int *__restrict__ lut, const float4 *__restrict__ frec,
                                                   const unsigned *__restrict__ fcol, const uint2 *__restrict__ frow,
                                                   int final_w, int final_h, int off_x, int off_y,
                                                   uint8_t *__restrict__ out, const uint8_t *__restrict__ center,
                                                   int center_h, int center_w, int row_begin, int row_count, WarpStrides st,
                                                   int *status) {
    {
        const long long pair = blockIdx.z;      // pair of a batch: its own image, canvas, inverses and records
        img += pair * st.img;
        out += pair * st.out;
        if (kBlend) center += pair * st.center;
        hinv_pad += pair * st.hinv;
        frec += pair * st.frec;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j0 = ((int)blockIdx.x * 64 + lane) * 4;
    const int y_first = row_begin + ((int)blockIdx.y * (int)(blockDim.x >> 6) + wave) * kRows;
    const int y_end = min(y_first + kRows, row_begin + row_count);
    if (j0 >= final_w || y_first >= y_end) return;
    const unsigned last = (unsigned)img_h * (unsigned)img_w * 3u - 4u;
    const unsigned clast = kBlend ? (unsigned)center_h * (unsigned)center_w * 3u - 4u : 0u;
    const int npx = min(4, final_w - j0);
    // column entries of the lane's four pixels (the table is padded to a multiple of 4 columns)
    const uint4 ce = *reinterpret_cast<const uint4 *>(fcol + j0);
    const unsigned cev[4] = {ce.x, ce.y, ce.z, ce.w};
    unsigned col[4];
    float dxf[4];
    // row entries: wave-uniform, scalar loads; each carries the 16-bit stamp of the sizes the tables were built for.
    // The tables' values become indices below: only if they were built for this mesh and canvas (warp_stamp16).  NOT an early
    // exit - hipcc then sinks the column load below the branch, one more round trip in every wave's chain (+0.35 us at C3,
    // measured) - and not a scalar load of its own (+0.25 us), but masks: on a foreign workspace every index becomes 0 (an
    // ordinary cell: in bounds), nothing is stored, and the status word says so.
    unsigned rr[kRows];
    float dyf[kRows];
#pragma unroll
    for (int t = 0; t < kRows; ++t) {
        const uint2 e = frow[(unsigned)min(y_first + t, y_end - 1)];
        rr[t] = __builtin_amdgcn_readfirstlane(e.x);
        dyf[t] = __uint_as_float(__builtin_amdgcn_readfirstlane(e.y));
    }
    const bool ready = (rr[0] >> 16) == warp_stamp16(mesh_rows, mesh_cols, final_w, final_h);
    const unsigned cmask = ready ? 0xffffu : 0u;
    if (!ready && lane == 0) atomicOr(status, apap::kStatusUnprepared);
#pragma unroll
    for (int t = 0; t < kRows; ++t) rr[t] &= cmask;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        col[k] = cev[k] & cmask;
        dxf[k] = (float)((cev[k] >> 16) & 0xffu) - 128.0f;    // v_cvt_f32_ubyte2; the byte is biased by 128
    }
    unsigned off[kRows][4];
    unsigned int px[kRows][4];
    unsigned long long doubt[kRows][4];    // lane masks (scalar registers)
    const unsigned rec_stride = (unsigned)(mesh_cols + 1);
    unsigned todo = (1u << kRows) - 1u;
    while (todo != 0u) {    // one pass per cell row the strip touches; wave-uniform
        const int first = __builtin_ctz(todo);
        unsigned r = rr[0];
#pragma unroll
        for (int t = 1; t < kRows; ++t) r = (t == first) ? rr[t] : r;
        const unsigned base = r * rec_stride;
        const float4 *pa = frec + (size_t)(base + col[0]) * 3, *pb = frec + (size_t)(base + col[3]) * 3;
        const float4 a0 = pa[0], a1 = pa[1], a2 = pa[2], b0 = pb[0], b1 = pb[1], b2 = pb[2];
        // per pixel: x-dependent parts of the three sums, the y coefficients, anchor, window
        float nx0[4], ny0[4], dn0[4], bx[4], by[4], h7[4];
        int n0x[4], n0y[4];
        unsigned thr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool is_a = (k == 0) || (k != 3 && col[k] == col[0]);
            const float4 q0 = is_a ? a0 : b0, q1 = is_a ? a1 : b1, q2 = is_a ? a2 : b2;
            nx0[k] = __builtin_fmaf(q0.x, dxf[k], q0.z);
            ny0[k] = __builtin_fmaf(q0.w, dxf[k], q1.y);
            dn0[k] = __builtin_fmaf(q1.w, dxf[k], q1.z);
            bx[k] = q0.y; by[k] = q1.x; h7[k] = q2.x;
            n0x[k] = __float_as_int(q2.y); n0y[k] = __float_as_int(q2.z);
            thr[k] = __float_as_uint(q2.w);
            // a third cell inside four pixels (cells narrower than the group): exact path
            if (k == 1 || k == 2) thr[k] = (!is_a && col[k] != col[3]) ? 0xffffffffu : thr[k];
        }
#pragma unroll
        for (int t = 0; t < kRows; ++t) {
            if (rr[t] != r) continue;
            todo &= ~(1u << t);
            const float dy = dyf[t];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float num_x = __builtin_fmaf(bx[k], dy, nx0[k]);
                const float num_y = __builtin_fmaf(by[k], dy, ny0[k]);
                const float den = __builtin_fmaf(h7[k], dy, dn0[k]);
                const float rc = __builtin_amdgcn_rcpf(den);
                const int fx = floor_to_int(num_x * rc), fy = floor_to_int(num_y * rc);      // 16.16 fixed point; NaN -> 0
                // integer half + anchor, fraction halves: v_add_u32_sdwa (sext WORD_1), v_min_u32_sdwa (WORD_0, WORD_0)
                const int ix = n0x[k] + (fx >> kFastFracBits), iy = n0y[k] + (fy >> kFastFracBits);
                const unsigned lo = min((unsigned)fx & 0xffffu, (unsigned)fy & 0xffffu);
                const bool ok = ((unsigned)ix < (unsigned)img_w) & ((unsigned)iy < (unsigned)img_h);
                off[t][k] = ok ? (__umul24((unsigned)iy, (unsigned)img_w) + (unsigned)ix) * 3u : 0xffffffffu;
                doubt[t][k] = __builtin_amdgcn_ballot_w64(lo < thr[k]);
            }
        }
    }
    // pixels in doubt: the exact float64 sequence.  Only waves that hold one come here.
    {
        unsigned long long any = 0;
#pragma unroll
        for (int t = 0; t < kRows; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) any |= doubt[t][k];
        if (any != 0) {
            unsigned bits = 0;
#pragma unroll
            for (int t = 0; t < kRows; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k) bits |= ((doubt[t][k] >> lane) & 1ull) ? (1u << (t * 4 + k)) : 0u;
            while (bits != 0u) {
                const int idx = __builtin_ctz(bits);
                bits &= bits - 1u;
                const int i = min(y_first + (idx >> 2), y_end - 1), j = min(j0 + (idx & 3), final_w - 1);
                unsigned cr = rr[0], cc = col[0];
#pragma unroll
                for (int t = 1; t < kRows; ++t) cr = (idx >> 2) == t ? rr[t] : cr;
#pragma unroll
                for (int k = 1; k < 4; ++k) cc = (idx & 3) == k ? col[k] : cc;
                unsigned o = exact_offset(hinv_pad, lut, mesh_rows, mesh_cols, final_h, cr, cc, i, j, off_x, off_y, img_w, img_h);
#pragma unroll
                for (int t = 0; t < kRows; ++t)
#pragma unroll
                    for (int k = 0; k < 4; ++k) off[t][k] = (idx == t * 4 + k) ? o : off[t][k];
            }
        }
    }
    // all of the strip's gathers in flight together (issuing a row's as soon as its offsets exist, or dropping
    // the range tests and the last-pixel guard for waves wholly inside the source, measured no faster: DESIGN.md;
    // one 16-byte window load per lane and row with the gathers kept for the lanes it cannot serve, and two 12-byte
    // windows per lane and row for every lane: byte-identical and 13-26 % slower, profiles/r05_k3_experiments.txt 2b, 2c)
#pragma unroll
    for (int t = 0; t < kRows; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) px[t][k] = gather_px(img, off[t][k], last);
    const int y_store_end = ready ? y_end : y_first;     // a foreign workspace: no row is stored
#pragma unroll
    for (int t = 0; t < kRows; ++t) {
        const int y = y_first + t;
        if (y >= y_store_end) break;  // wave-uniform
        if (kBlend) {
            const int ci = y - off_y;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cj = j0 + k - off_x;
                const bool in = ci >= 0 && ci < center_h && cj >= 0 && cj < center_w;
                const unsigned co = in ? ((unsigned)ci * (unsigned)center_w + (unsigned)cj) * 3u : 0u;
                const unsigned cc = co < clast ? co : clast;
                unsigned int c;
                __builtin_memcpy(&c, center + cc, 4);
                c = in ? ((c >> (8 * (co - cc))) & 0x00ffffffu) : 0u;
                const unsigned w = px[t][k];
                const unsigned avg = (w & c) + (((w ^ c) & 0x00fefefeu) >> 1);
                px[t][k] = (w != 0u && c != 0u) ? avg : (w | c);
            }
        }
        uint8_t *o = out + ((size_t)(y - row_begin) * (size_t)final_w) * 3 + (unsigned)j0 * 3u;
        if (npx == 4) {
            // 4 x 24 bits -> 3 dwords (v_perm_b32 picks bytes 0-3 from its second operand, 4-7 from its first), stored
            // NON-TEMPORAL: the canvas is written once and never read back by this kernel - the bytes stream past the L2
            // instead of waiting in it, dirty, for the write-back at the end of the kernel (tools/k3_policy.hip: of the
            // eight sc0 / sc1 / nt combinations on the stores and the eight on the gathers, nt stores + plain loads is the
            // fastest; K3 at C3 16.0 -> 15.1 us warm, 19.3 -> 17.5 us cold)
            store12_stream(o, px[t][0] | (px[t][1] << 24), __builtin_amdgcn_perm(px[t][2], px[t][1], 0x05040201u),
                           __builtin_amdgcn_perm(px[t][3], px[t][2], 0x06050402u));
        } else {
            for (int k = 0; k < npx; ++k) {
                o[3 * k] = (uint8_t)(px[t][k] & 0xff);
                o[3 * k + 1] = (uint8_t)((px[t][k] >> 8) & 0xff);
                o[3 * k + 2] = (uint8_t)((px[t][k] >> 16) & 0xff);
            }
        }
    }
}

// coordinates only (parity tests of the arithmetic of k_warp)
__global__ __launch_bounds__(256) void k_warp_coords(const double *__restrict__ hinv_pad, int mesh_cols,
                                                     const int *__restrict__ lut, int final_w,
                                                     int final_h, int off_x, int off_y,
                                                     double *__restrict__ coords) {
    const size_t total = (size_t)final_w * final_h;
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int i = (int)(g / final_w);
    const int j = (int)(g - (size_t)i * final_w);
    const int cell = lut[i] * mesh_cols + lut[final_h + j];
    double tx, ty;
    target_of(hinv_pad, cell, (double)(j - off_x), (double)(i - off_y), tx, ty);
    coords[2 * g] = tx;
    coords[2 * g + 1] = ty;
}

// output stage, apap.py:250-264
__global__ __launch_bounds__(256) void k_flatten(const float *__restrict__ H, int cells,
                                                 double *__restrict__ out, int *status) {
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= cells) return;
    double m[9], r[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) m[k] = (double)H[(size_t)cell * 9 + k];
    if (!inv3(m, r)) atomicOr(status, apap::kStatusSingular);
    float f[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) f[k] = (float)r[k];
    const float d = f[8];
#pragma unroll
    for (int k = 0; k < 9; ++k) f[k] = f[k] / d;  // float32 division like numpy's in-place /=
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(size_t)cell * 9 + 3 * c + rr] = (double)f[3 * rr + c];
}

// uniform_blend, apap_utils.py:75-88
__global__ __launch_bounds__(256) void k_blend(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
                                               size_t pixels, uint8_t *__restrict__ out) {
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < pixels;
         p += (size_t)gridDim.x * blockDim.x) {
        const uint8_t *pa = a + 3 * p, *pb = b + 3 * p;
        const unsigned a0 = pa[0], a1 = pa[1], a2 = pa[2], b0 = pb[0], b1 = pb[1], b2 = pb[2];
        const bool both = (a0 + a1 + a2 > 0) && (b0 + b1 + b2 > 0);
        const unsigned sh = both ? 1u : 0u;
        // float64 sum times 0.5 (both non-black) or 1, then astype(uint8) = truncation.
        // When the factor is 1 one side is all zero, so the sum never exceeds 255.
        out[3 * p] = (uint8_t)((a0 + b0) >> sh);
        out[3 * p + 1] = (uint8_t)((a1 + b1) >> sh);
        out[3 * p + 2] = (uint8_t)((a2 + b2) >> sh);
    }
}

inline int hip_fail(hipError_t e, const char *what) { return apap::hip_fail((int)e, what); }
using apap::ProfScope;

}  // namespace

namespace apap {

int hip_fail(int hip_error, const char *what) {
    return fail(APAP_ERR_HIP, "%s: %s", what, hipGetErrorString((hipError_t)hip_error));
}

int opt(const apap_ctx *ctx, int which) {
    static const apap_ctx defaults;      // never written
    return (ctx ? ctx : &defaults)->opt[which];
}

ProfScope::ProfScope(apap_ctx *ctx, int slot, void *stream)
    : ctx_(ctx), stream_(stream), a_(nullptr), b_(nullptr), slot_(slot), on_(ctx && ctx->opt[APAP_OPT_PROFILE]) {
    if (!on_) return;
    hipEvent_t a, b;
    on_ = hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess;
    if (!on_) return;
    a_ = a;
    b_ = b;
    (void)hipEventRecord(a, (hipStream_t)stream_);
}

ProfScope::~ProfScope() {
    if (!on_) return;
    (void)hipEventRecord((hipEvent_t)b_, (hipStream_t)stream_);
    ctx_->spans.push_back(ProfSpan{slot_, a_, b_});
}

SolvePlan plan_solve(int n, int cells, int variant, int batch, int want_waves, int plan_cells, int moments) {
    SolvePlan p{};
    if (variant == APAP_VARIANT_AUTO) variant = APAP_VARIANT_MFMA;  // measured faster on C2-C4, DESIGN.md
    p.variant = variant;
    p.cell_tiles = (cells + kWave - 1) / kWave;  // both variants: 64 cells per 4-wave block
    p.cells_pad = p.cell_tiles * kWave;
    // Fill the chip (256 CUs x 4 SIMDs): aim at >= 4 waves per SIMD (also evens out the
    // blocks-per-CU imbalance: 2500 waves leave SIMDs with 2 or 3, 5000 with 4 or 5).  Small meshes split
    // the keypoint list over grid.y; each split writes its own moment slab and K2 adds
    // the slabs in a fixed order.  A split is at least one LDS chunk of keypoints and a whole
    // number of chunks (a partial chunk costs as much as a full one in the MFMA kernel).
    int splits = 1;
    if (want_waves < 1) want_waves = 4096;
    // waves per (64-cell tile, split): the MFMA kernels' block is 4 waves on one tile (two tiles for
    // APAP_VARIANT_MFMA4X2), the VALU kernel's is 4 waves on 4 tiles
    const int waves_per_2tiles = variant == APAP_VARIANT_MFMA || variant == APAP_VARIANT_MFMA4 ? 8
                                 : variant == APAP_VARIANT_MFMA4X2 ? 4 : 2;
    // APAP_OPT_PLAN_CELLS: the splits one pair of that many cells would get, whatever this launch holds
    const long long plan_tiles = plan_cells > 0 ? (plan_cells + kWave - 1) / kWave : (long long)p.cell_tiles * batch;
    while (splits < 32 && plan_tiles * waves_per_2tiles * splits < 2 * want_waves && n / (splits * 2) >= kChunk) splits *= 2;
    int pps = (n + splits - 1) / splits;
    pps = (pps + kChunk - 1) / kChunk * kChunk;
    p.splits = (n + pps - 1) / pps;  // no empty split
    p.pts_per_split = pps;
    p.moment_bytes = (size_t)p.splits * (moments == 24 ? 24 : kMoments) * p.cells_pad * sizeof(double);
    return p;
}

}  // namespace apap

extern "C" {

apap_ctx *apap_ctx_create(void) { return new (std::nothrow) apap_ctx; }

void apap_ctx_destroy(apap_ctx *ctx) {
    if (!ctx) return;
    for (apap::ProfSpan &sp : ctx->spans) {
        (void)hipEventDestroy((hipEvent_t)sp.a);
        (void)hipEventDestroy((hipEvent_t)sp.b);
    }
    for (apap::DevSlot &sl : ctx->slots)
        if (sl.ptr) (void)hipFree(sl.ptr);
    for (void *e : ctx->events) (void)hipEventDestroy((hipEvent_t)e);
    for (void *st : ctx->streams)
        if (st) (void)hipStreamDestroy((hipStream_t)st);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    delete ctx;
}

int apap_ctx_set_option(apap_ctx *ctx, int option, int value) {
    if (!ctx) return apap::fail(APAP_ERR_INVALID_ARG, "apap_ctx_set_option: null context (NULL means the defaults and cannot be changed)");
    bool ok = false;
    switch (option) {
        case APAP_OPT_SOLVER_VARIANT: ok = value >= APAP_VARIANT_AUTO && value <= APAP_VARIANT_MFMA4X2; break;
        case APAP_OPT_EIGEN_SOLVER: ok = value >= APAP_EIGEN_AUTO && value <= APAP_EIGEN_INVERSE_ITERATION; break;
        case APAP_OPT_CAREFUL:
        case APAP_OPT_PROFILE: ok = value == 0 || value == 1; break;
        case APAP_OPT_WANT_WAVES: ok = value >= 1; break;
        case APAP_OPT_WARP_ROWS: ok = value == 0 || value == 1 || value == 2 || value == 4 || value == 5 || value == 6 || value == 8; break;
        case APAP_OPT_WEIGHT_CHUNK_KB: ok = value >= 1; break;
        case APAP_OPT_FUSED_MAX_CELLS:
        case APAP_OPT_PLAN_CELLS: ok = value >= 0; break;
        case APAP_OPT_WARP_FAST:
        case APAP_OPT_OVERLAP_PCIE:
        case APAP_OPT_WEIGHTS_F32: ok = value == 0 || value == 1; break;
        case APAP_OPT_MOMENTS: ok = value == 30 || value == 24; break;
        default: return apap::fail(APAP_ERR_INVALID_ARG, "apap_ctx_set_option: unknown option %d", option);
    }
    if (!ok) return apap::fail(APAP_ERR_INVALID_ARG, "apap_ctx_set_option: value %d is not valid for option %d", value, option);
    ctx->opt[option] = value;
    return APAP_OK;
}

int apap_ctx_get_option(const apap_ctx *ctx, int option, int *value) {
    if (!value || option < 0 || option >= APAP_OPT_COUNT) return apap::fail(APAP_ERR_INVALID_ARG, "apap_ctx_get_option: bad argument");
    *value = apap::opt(ctx, option);
    return APAP_OK;
}

int apap_ctx_profile_read(apap_ctx *ctx, float *ms, int *launches) {
    if (!ctx || !ms || !launches) return apap::fail(APAP_ERR_INVALID_ARG, "apap_ctx_profile_read: null argument");
    for (int k = 0; k < APAP_PROF_SLOTS; ++k) {
        ms[k] = 0.0f;
        launches[k] = 0;
    }
    int rc = APAP_OK;
    for (apap::ProfSpan &sp : ctx->spans) {
        float t = 0.0f;
        hipError_t e = hipEventSynchronize((hipEvent_t)sp.b);
        if (e == hipSuccess) e = hipEventElapsedTime(&t, (hipEvent_t)sp.a, (hipEvent_t)sp.b);
        if (e != hipSuccess) rc = hip_fail(e, "apap_ctx_profile_read");
        ms[sp.slot] += t;
        launches[sp.slot] += 1;
        (void)hipEventDestroy((hipEvent_t)sp.a);
        (void)hipEventDestroy((hipEvent_t)sp.b);
    }
    ctx->spans.clear();
    return rc;
}

size_t apap_solve_batch_workspace_bytes(apap_ctx *ctx, int n, int cells, int batch) {
    if (n < 1 || cells < 1 || batch < 1) return 0;
    return apap::plan_solve(n, cells, apap::opt(ctx, APAP_OPT_SOLVER_VARIANT), batch, apap::opt(ctx, APAP_OPT_WANT_WAVES), apap::opt(ctx, APAP_OPT_PLAN_CELLS),
                            apap::opt(ctx, APAP_OPT_MOMENTS)).moment_bytes * (size_t)batch;
}

size_t apap_solve_workspace_bytes(apap_ctx *ctx, int n, int cells) { return apap_solve_batch_workspace_bytes(ctx, n, cells, 1); }

}  // extern "C"

static int solve_batch_impl(apap_ctx *ctx, const double *d_tables, int n, const double *d_vertices, long long vertices_stride,
                            int cells, double gamma, double sigma, const double *d_denorms, float *d_H,
                            int batch, void *d_work, size_t work_bytes, void *stream, const WarpEmit &we) {
    if (!d_tables || !d_vertices || !d_denorms || !d_H || !d_work)
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_solve_device: null device pointer");
    if (n < 1 || cells < 1 || batch < 1 || batch > 65535 || vertices_stride < 0)
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_solve_device: n=%d cells=%d batch=%d", n, cells, batch);
    const bool m24 = apap::opt(ctx, APAP_OPT_MOMENTS) == 24;
    const bool w32 = m24 && apap::opt(ctx, APAP_OPT_WEIGHTS_F32) != 0;   // float32 weights exist in the 24-sum kernels only
    if (m24 && apap::opt(ctx, APAP_OPT_SOLVER_VARIANT) == APAP_VARIANT_VALU)
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_solve_device: APAP_OPT_MOMENTS = 24 has no VALU form (use AUTO, MFMA, MFMA4 or MFMA4X2)");
    const apap::SolvePlan p = apap::plan_solve(n, cells, apap::opt(ctx, APAP_OPT_SOLVER_VARIANT), batch, apap::opt(ctx, APAP_OPT_WANT_WAVES),
                                               apap::opt(ctx, APAP_OPT_PLAN_CELLS), m24 ? 24 : 30);
    if (work_bytes < p.moment_bytes * (size_t)batch)
        return apap::fail(APAP_ERR_WORKSPACE, "apap_solve_device: workspace %zu < %zu bytes", work_bytes,
                          p.moment_bytes * (size_t)batch);
    hipStream_t s = (hipStream_t)stream;
    const double inv_sigma = 1.0 / (sigma * sigma);  // apap.py:142
    const double inv_sigma2 = 2.0 * inv_sigma;       // K1 evaluates exp(-2 d / sigma^2) = w^2
    const double gamma2 = gamma > 0.0 ? gamma * gamma : 0.0;
    double *moments = (double *)d_work;
    BatchStride bs;
    bs.table = (long long)n * APAP_TABLE_STRIDE;
    bs.vertices = vertices_stride;
    bs.moments = (long long)(p.moment_bytes / sizeof(double));
    bs.denorm = APAP_DENORM_DOUBLES;
    bs.H = (long long)cells * 9;
    const int pick_rank = 9 - (2 * n < 9 ? 2 * n : 9);
    const int careful = apap::opt(ctx, APAP_OPT_CAREFUL);
    // Small meshes (AUTO only): one fused launch, 16 cells per block.  Up to 4096 cells in all the
    // blocks fit the chip in about one round at the fused kernel's one wave per SIMD.
    const int plan_cells = apap::opt(ctx, APAP_OPT_PLAN_CELLS);
    if (!m24 && apap::opt(ctx, APAP_OPT_SOLVER_VARIANT) == APAP_VARIANT_AUTO &&    // (the fused kernel reads the 30-sum table)
        (plan_cells > 0 ? (long long)plan_cells : (long long)cells * batch) <= apap::opt(ctx, APAP_OPT_FUSED_MAX_CELLS)) {
        ProfScope prof(ctx, APAP_PROF_ASSEMBLE, s);   // reported under the K1 slot; the K2 slot stays empty
        const dim3 grid((cells + 15) / 16, 1, batch);
        if (apap::opt(ctx, APAP_OPT_EIGEN_SOLVER) == APAP_EIGEN_JACOBI)
            hipLaunchKernelGGL(k_solve_small<false>, grid, dim3(kSmallThreads), 0, s, d_tables, n, d_vertices, cells, gamma, inv_sigma,
                               d_denorms, pick_rank, careful, d_H, bs, we);
        else
            hipLaunchKernelGGL(k_solve_small<true>, grid, dim3(kSmallThreads), 0, s, d_tables, n, d_vertices, cells, gamma, inv_sigma,
                               d_denorms, pick_rank, careful, d_H, bs, we);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "apap_solve_device launch");
        return APAP_OK;
    }
    {
        ProfScope prof(ctx, APAP_PROF_ASSEMBLE, s);
        const dim3 grid(p.cell_tiles, p.splits, batch);
        // 4 waves (64 cells) per block measured best: 2 -> 224 us, 4 -> 201 us, 8 -> 227 us at C3
#define APAP_K1_ARGS d_tables, n, d_vertices, cells, p.cells_pad, gamma2, inv_sigma2, p.pts_per_split, moments, bs
        const dim3 grid2((p.cell_tiles + 1) / 2, p.splits, batch);   // two 64-cell tiles per block
        if (p.variant == APAP_VARIANT_MFMA) {
            if (!m24) hipLaunchKernelGGL((k_assemble_mfma<4, false, false>), grid, dim3(256), 0, s, APAP_K1_ARGS);
            else if (!w32) hipLaunchKernelGGL((k_assemble_mfma<4, true, false>), grid, dim3(256), 0, s, APAP_K1_ARGS);
            else hipLaunchKernelGGL((k_assemble_mfma<4, true, true>), grid, dim3(256), 0, s, APAP_K1_ARGS);
        } else if (p.variant == APAP_VARIANT_MFMA4) {
            if (!m24) hipLaunchKernelGGL((k_assemble_mfma4<1, 8, false>), grid, dim3(256), 0, s, APAP_K1_ARGS);
            else if (!w32) hipLaunchKernelGGL((k_assemble_mfma4<1, 6, false>), grid, dim3(256), 0, s, APAP_K1_ARGS);
            else hipLaunchKernelGGL((k_assemble_mfma4<1, 6, true>), grid, dim3(256), 0, s, APAP_K1_ARGS);
        } else if (p.variant == APAP_VARIANT_MFMA4X2) {
            if (!m24) hipLaunchKernelGGL((k_assemble_mfma4<2, 8, false>), grid2, dim3(256), 0, s, APAP_K1_ARGS);
            else if (!w32) hipLaunchKernelGGL((k_assemble_mfma4<2, 6, false>), grid2, dim3(256), 0, s, APAP_K1_ARGS);
            else hipLaunchKernelGGL((k_assemble_mfma4<2, 6, true>), grid2, dim3(256), 0, s, APAP_K1_ARGS);
        } else {
            hipLaunchKernelGGL(k_assemble_valu, dim3((p.cell_tiles + 3) / 4, p.splits, batch), dim3(256), 0, s, APAP_K1_ARGS);
        }
#undef APAP_K1_ARGS
    }
    {
        ProfScope prof(ctx, APAP_PROF_EIGEN, s);
        const dim3 grid(p.cell_tiles, 1, batch);
#define APAP_K2_ARGS moments, p.splits, cells, p.cells_pad, d_denorms, pick_rank, d_H, bs, d_tables, n, d_vertices, gamma, inv_sigma, careful, we
        const bool jacobi = apap::opt(ctx, APAP_OPT_EIGEN_SOLVER) == APAP_EIGEN_JACOBI;
        if (jacobi && m24) hipLaunchKernelGGL((k_eigen_denorm<false, true>), grid, dim3(64), 0, s, APAP_K2_ARGS);
        else if (jacobi) hipLaunchKernelGGL((k_eigen_denorm<false, false>), grid, dim3(64), 0, s, APAP_K2_ARGS);
        else if (m24) hipLaunchKernelGGL((k_eigen_denorm<true, true>), grid, dim3(64), 0, s, APAP_K2_ARGS);
        else hipLaunchKernelGGL((k_eigen_denorm<true, false>), grid, dim3(64), 0, s, APAP_K2_ARGS);
#undef APAP_K2_ARGS
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "apap_solve_device launch");
    return APAP_OK;
}

extern "C" {

int apap_solve_batch_device(apap_ctx *ctx, const double *d_tables, int n, const double *d_vertices, long long vertices_stride,
                            int cells, double gamma, double sigma, const double *d_denorms, float *d_H,
                            int batch, void *d_work, size_t work_bytes, void *stream) {
    const WarpEmit none{};
    return solve_batch_impl(ctx, d_tables, n, d_vertices, vertices_stride, cells, gamma, sigma, d_denorms, d_H, batch, d_work,
                            work_bytes, stream, none);
}

int apap_solve_warp_batch_device(apap_ctx *ctx, const double *d_tables, int n, const double *d_vertices, long long vertices_stride,
                                 double gamma, double sigma, const double *d_denorms, float *d_H, int batch, void *d_work,
                                 size_t work_bytes, int mesh_rows, int mesh_cols, const double *d_mesh_w, int n_w,
                                 const double *d_mesh_h, int n_h, int final_w, int final_h, int off_x, int off_y,
                                 void *d_warp_work, size_t warp_work_bytes, int *d_status, void *stream) {
    if (!d_mesh_w || !d_mesh_h || !d_warp_work || !d_status)
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_solve_warp_batch_device: null device pointer");
    if (mesh_rows < 1 || mesh_cols < 1 || n_w < 1 || n_h < 1 || final_w < 1 || final_h < 1 || batch < 1)
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_solve_warp_batch_device: bad size");
    if (n_w > kMaxEdges || n_h > kMaxEdges || mesh_rows >= 65535 || mesh_cols >= 65535 ||
        (unsigned long long)mesh_rows * (unsigned long long)mesh_cols * APAP_HINV_STRIDE * sizeof(double) >= (1ull << 32))
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_solve_warp_batch_device: mesh too large for the warp's lookup tables (solve and "
                                                "warp it with the separate entry points)");
    const size_t need = apap_warp_batch_workspace_bytes(mesh_rows, mesh_cols, final_w, final_h, batch);
    if (warp_work_bytes < need)
        return apap::fail(APAP_ERR_WORKSPACE, "apap_solve_warp_batch_device: warp workspace %zu < %zu bytes", warp_work_bytes, need);
    const WarpWork ww = warp_work_layout(d_warp_work, mesh_rows, mesh_cols, final_w, final_h, batch);
    WarpEmit we;
    we.hinv_pad = ww.hinv_pad; we.frec = ww.frec;
    we.hinv_stride = ww.hinv_stride; we.frec_stride = ww.frec_stride;
    we.mesh_w = d_mesh_w; we.mesh_h = d_mesh_h; we.n_w = n_w; we.n_h = n_h;
    we.mesh_rows = mesh_rows; we.mesh_cols = mesh_cols; we.final_w = final_w; we.final_h = final_h; we.off_x = off_x; we.off_y = off_y;
    we.status = d_status;
    return solve_batch_impl(ctx, d_tables, n, d_vertices, vertices_stride, mesh_rows * mesh_cols, gamma, sigma, d_denorms, d_H, batch,
                            d_work, work_bytes, stream, we);
}

int apap_solve_device(apap_ctx *ctx, const double *d_table, int n, const double *d_vertices, int cells,
                      double gamma, double sigma, const double *d_denorm, float *d_H, void *d_work,
                      size_t work_bytes, void *stream) {
    return apap_solve_batch_device(ctx, d_table, n, d_vertices, 0, cells, gamma, sigma, d_denorm, d_H, 1, d_work, work_bytes,
                                   stream);
}

int apap_weights_device(apap_ctx *ctx, const double *d_table, int n, const double *d_vertices, int cells,
                        double gamma, double sigma, double *d_W, void *stream) {
    if (!d_table || !d_vertices || !d_W) return apap::fail(APAP_ERR_INVALID_ARG, "apap_weights_device: null device pointer");
    if (n < 1 || cells < 1) return apap::fail(APAP_ERR_INVALID_ARG, "apap_weights_device: n=%d cells=%d", n, cells);
    const size_t total = (size_t)cells * n;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_weights, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_table, n, d_vertices, cells,
                       gamma, 1.0 / (sigma * sigma), d_W);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "apap_weights_device launch");
    return APAP_OK;
}

size_t apap_warp_batch_workspace_bytes(int mesh_rows, int mesh_cols, int final_w, int final_h, int batch) {
    if (mesh_rows < 1 || mesh_cols < 1 || final_w < 1 || final_h < 1 || batch < 1) return 0;
    return warp_work_layout(nullptr, mesh_rows, mesh_cols, final_w, final_h, batch).bytes;
}

size_t apap_warp_workspace_bytes(int mesh_rows, int mesh_cols, int final_w, int final_h) {
    return apap_warp_batch_workspace_bytes(mesh_rows, mesh_cols, final_w, final_h, 1);
}

}  // extern "C"

namespace {

// Everything a warp launch needs; the entry points below fill it in.  `batch` pairs share the mesh edges, the canvas
// size and the offsets; pair k reads image img + k img_stride (bytes), grid Hfwd + k cells 9, and writes canvas
// out + k out_stride (bytes) - out points at row `row_begin` of pair 0's canvas.
struct WarpArgs {
    const uint8_t *img = nullptr;
    long long img_stride = 0;
    int img_h = 0, img_w = 0;
    const uint8_t *center = nullptr;
    long long center_stride = 0;
    int center_h = 0, center_w = 0;
    const void *Hfwd = nullptr;
    int mesh_rows = 0, mesh_cols = 0;
    const double *mesh_w = nullptr;
    int n_w = 0;
    const double *mesh_h = nullptr;
    int n_h = 0;
    int final_w = 0, final_h = 0, off_x = 0, off_y = 0;
    uint8_t *out = nullptr;
    long long out_stride = 0;
    void *Hinv_out = nullptr;
    int batch = 1;
    void *work = nullptr;
    size_t work_bytes = 0;
    int *status = nullptr;
    void *stream = nullptr;
    int row_begin = 0, row_count = 0;
    int phase = APAP_WARP_ALL;
    int **src_rows = nullptr;
};

template <typename T>
int warp_prologue(apap_ctx *ctx, const WarpArgs &a, WarpWork *ww, bool *fast_tables) {
    if (!a.mesh_w || !a.mesh_h || !a.work || !a.status) return apap::fail(APAP_ERR_INVALID_ARG, "warp: null device pointer");
    if (a.mesh_rows < 1 || a.mesh_cols < 1 || a.n_w < 1 || a.n_h < 1 || a.final_w < 1 || a.final_h < 1 || a.batch < 1 || a.batch > 65535)
        return apap::fail(APAP_ERR_INVALID_ARG, "warp: bad size");
    const size_t need = apap_warp_batch_workspace_bytes(a.mesh_rows, a.mesh_cols, a.final_w, a.final_h, a.batch);
    if (a.work_bytes < need) return apap::fail(APAP_ERR_WORKSPACE, "warp: workspace %zu < %zu bytes", a.work_bytes, need);
    const int cells = a.mesh_rows * a.mesh_cols;
    *ww = warp_work_layout(a.work, a.mesh_rows, a.mesh_cols, a.final_w, a.final_h, a.batch);
    *fast_tables = a.n_w <= kMaxEdges && a.n_h <= kMaxEdges && a.mesh_rows < 65535 && a.mesh_cols < 65535;
    const bool geometry = (a.phase & APAP_WARP_GEOMETRY) != 0, per_cell = (a.phase & APAP_WARP_CELLS) != 0;
    if (!geometry && !per_cell) return APAP_OK;     // the tables an earlier call left in this workspace are used
    if (per_cell && !a.Hfwd) return apap::fail(APAP_ERR_INVALID_ARG, "warp: null H grid");
    hipStream_t s = (hipStream_t)a.stream;
    const T *Hfwd = (const T *)a.Hfwd;
    T *Hinv_out = (T *)a.Hinv_out;
    if (*fast_tables) {
        // one launch: cell inverses + fast records (grid.y = pair) + lookup tables (reported under the INVERT slot)
        ProfScope prof(ctx, APAP_PROF_INVERT, s);
        const int inv_blocks = per_cell ? (int)(((size_t)(a.mesh_rows + 1) * (a.mesh_cols + 1) + 255) / 256) : 0;
        const int lut_blocks = geometry ? (a.final_h + 1023) / 1024 + (a.final_w + 1023) / 1024 : 0;
        hipLaunchKernelGGL(k_warp_setup<T>, dim3(inv_blocks + lut_blocks, per_cell ? a.batch : 1), dim3(256), 0, s, Hfwd, cells,
                           ww->hinv_pad, Hinv_out, inv_blocks, a.mesh_w, a.n_w, a.mesh_h, a.n_h, a.mesh_rows, a.mesh_cols,
                           a.final_w, a.final_h, ww->lut, a.status, a.off_x, a.off_y, ww->frec, ww->fcol, ww->frow,
                           a.src_rows && geometry && per_cell && a.batch == 1 ? ww->src_rows : (int *)nullptr, ww->hinv_stride,
                           ww->frec_stride);
    } else {
        if (per_cell) {
            ProfScope prof(ctx, APAP_PROF_INVERT, s);
            for (int k = 0; k < a.batch; ++k)
                hipLaunchKernelGGL(k_invert_cells<T>, dim3((cells + 255) / 256), dim3(256), 0, s, Hfwd + (size_t)k * cells * 9, cells,
                                   ww->hinv_pad + k * ww->hinv_stride, Hinv_out ? Hinv_out + (size_t)k * cells * 9 : (T *)nullptr,
                                   a.status);
        }
        if (geometry) {
            ProfScope prof(ctx, APAP_PROF_LUT, s);
            hipLaunchKernelGGL(k_cell_lut, dim3((a.final_w + a.final_h + 255) / 256), dim3(256), 0, s, a.mesh_w, a.n_w,
                               a.mesh_h, a.n_h, a.mesh_rows, a.mesh_cols, a.final_w, a.final_h, ww->lut, a.status);
        }
    }
    return APAP_OK;
}

template <typename T>
int warp_impl(apap_ctx *ctx, const WarpArgs &a) {
    const bool gather = (a.phase & APAP_WARP_GATHER) != 0;
    if (a.phase < 1 || a.phase > APAP_WARP_ALL) return apap::fail(APAP_ERR_INVALID_ARG, "warp: phases %d", a.phase);
    if (gather && (!a.img || !a.out)) return apap::fail(APAP_ERR_INVALID_ARG, "apap_warp_device: null image pointer");
    if (a.row_begin < 0 || a.row_count < 0 || (long long)a.row_begin + a.row_count > a.final_h)
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_warp_rows_device: rows [%d, %d) outside the canvas of %d rows",
                          a.row_begin, a.row_begin + a.row_count, a.final_h);
    if (gather && a.center) {
        // the reference pastes with a numpy slice assignment, which raises when the
        // centre image does not fit the canvas at the offsets
        if (a.center_h < 1 || a.center_w < 1 || (size_t)a.center_h * a.center_w < 2 || a.off_x < 0 || a.off_y < 0 ||
            (long long)a.off_y + a.center_h > a.final_h || (long long)a.off_x + a.center_w > a.final_w)
            return apap::fail(APAP_ERR_INVALID_ARG, "apap_stitch_device: centre image %dx%d at (%d,%d) does not fit canvas %dx%d",
                              a.center_w, a.center_h, a.off_x, a.off_y, a.final_w, a.final_h);
        if ((unsigned long long)a.center_h * (unsigned long long)a.center_w * 3ull >= (1ull << 32))
            return apap::fail(APAP_ERR_INVALID_ARG, "apap_stitch_device: centre image of 4 GiB or more");
    }
    if (gather) {
        if (a.img_h < 1 || a.img_w < 1 || (size_t)a.img_h * a.img_w < 2)
            return apap::fail(APAP_ERR_INVALID_ARG, "apap_warp_device: bad image size (need at least 2 pixels)");
        if ((unsigned long long)a.img_h * (unsigned long long)a.img_w * 3ull >= (1ull << 32))
            return apap::fail(APAP_ERR_INVALID_ARG, "apap_warp_device: image of 4 GiB or more");
    }
    if ((unsigned long long)a.final_w * (unsigned long long)a.final_h >= (1ull << 31))
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_warp_device: canvas of 2^31 pixels or more");
    if ((unsigned long long)a.mesh_rows * (unsigned long long)a.mesh_cols * APAP_HINV_STRIDE * sizeof(double) >= (1ull << 32))
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_warp_device: mesh of 53 million cells or more");
    hipStream_t s = (hipStream_t)a.stream;
    WarpWork ww;
    bool fast_tables;
    const int rc = warp_prologue<T>(ctx, a, &ww, &fast_tables);
    if (rc != APAP_OK) return rc;
    if (a.src_rows) *a.src_rows = fast_tables && a.batch == 1 ? ww.src_rows : nullptr;
    double *hinv_pad = ww.hinv_pad;
    int *lut = ww.lut;
    if (a.row_count == 0 || !gather) return APAP_OK;  // an empty band: only the set-up kernel ran
    const uint8_t *d_img = a.img, *d_center = a.center;
    uint8_t *d_out = a.out;
    const int img_h = a.img_h, img_w = a.img_w, center_h = a.center_h, center_w = a.center_w, mesh_rows = a.mesh_rows,
              mesh_cols = a.mesh_cols, final_w = a.final_w, final_h = a.final_h, off_x = a.off_x, off_y = a.off_y,
              row_begin = a.row_begin, row_count = a.row_count;
    const unsigned batch = (unsigned)a.batch;
    const WarpStrides st{a.img_stride, a.out_stride, a.center_stride, ww.hinv_stride, ww.frec_stride};
    const size_t total = (size_t)final_w * row_count;
    const size_t threads = (total + 3) / 4;
    // APAP_OPT_WARP_ROWS: 0 = flat-order kernel, 2 / 4 / 5 / 6 / 8 = row strips of that many rows per wave, 1 (default) =
    // chosen from the size of the launch.  A strip pays its table trips and its cell records once, so taller strips are
    // cheaper per pixel - but a launch needs a few generations of waves before that shows: measured on the final kernel
    // (profiles/r05_k3_experiments.txt 7), one C3 canvas (1.3 generations of 4-row strips) runs 14.9 us with 4 or 5 rows,
    // 15.3 with 6, 16.2 with 8; C4's canvas (5 generations) 45.1 / 43.9 / 43.7 / 44.6; 32 C5 pairs per launch 13.0 / 12.1 /
    // 12.7 / 11.9 us per pair; 8 C3 pairs 11.2 / 11.0 / 10.9 / 10.7.  The fused stitch keeps 4 rows (its taller forms need
    // more registers: one C3 stitch 32.8 us with 4 rows, 34.4 with 5, 37.2 with 6).
    int warp_kernel = apap::opt(ctx, APAP_OPT_WARP_ROWS);
    if (warp_kernel == 1) {
        const unsigned long long strips4 = (unsigned long long)batch * (unsigned long long)((final_w + 255) / 256) *
                                           (unsigned long long)((row_count + 3) / 4);      // waves of a 4-row launch
        warp_kernel = d_center ? 4 : strips4 >= 64000ull ? 8 : strips4 >= 24000ull ? 6 : 4;
    }
    // the strip kernel forms source offsets with 24-bit multiplies
    // ... and marks pixels outside the source with the sign bit of the byte offset
    const bool strips = warp_kernel > 0 && img_w < (1 << 24) && img_h < (1 << 24) &&
                        (unsigned long long)img_h * (unsigned long long)img_w * 3ull < (1ull << 31);
    // APAP_OPT_WARP_FAST (default 1): the float32-estimate kernel.  It is exact for any input, but a pixel the
    // estimate cannot decide costs a trip through the exact sequence, and every pixel of a cell wider than
    // 254 pixels is one: meshes that coarse (on average) keep the all-float64 strip kernel.
    const bool fast_ok = strips && fast_tables && apap::opt(ctx, APAP_OPT_WARP_FAST) && final_w / mesh_cols <= 128 &&
                         final_h / mesh_rows <= 128;
    if (fast_ok) {
        ProfScope prof(ctx, APAP_PROF_WARP, s);
        const int rows = warp_kernel >= 8 ? 8 : warp_kernel >= 4 ? warp_kernel : 2;       // instantiated for 2, 4, 5, 6, 8
        constexpr int kWpb = 256 / 64;      // waves (= strips) per block
        const dim3 grid((unsigned)((final_w + 255) / 256), (unsigned)((row_count + kWpb * rows - 1) / (kWpb * rows)), batch);
#define APAP_LAUNCH_FAST(R)                                                                                          \
    if (d_center)                                                                                                    \
        hipLaunchKernelGGL((k_warp_fast<true, R>), grid, dim3(256), 0, s, d_img, img_h, img_w, hinv_pad, mesh_rows, mesh_cols,  \
                           lut, ww.frec, ww.fcol, ww.frow, final_w, final_h, off_x, off_y, d_out, d_center,           \
                           center_h, center_w, row_begin, row_count, st, a.status);                                  \
    else                                                                                                             \
        hipLaunchKernelGGL((k_warp_fast<false, R>), grid, dim3(256), 0, s, d_img, img_h, img_w, hinv_pad, mesh_rows, mesh_cols, \
                           lut, ww.frec, ww.fcol, ww.frow, final_w, final_h, off_x, off_y, d_out,                     \
                           (const uint8_t *)nullptr, 0, 0, row_begin, row_count, st, a.status)
        if (rows == 4) { APAP_LAUNCH_FAST(4); }
        else if (rows == 5) { APAP_LAUNCH_FAST(5); }
        else if (rows == 6) { APAP_LAUNCH_FAST(6); }
        else if (rows >= 7) { APAP_LAUNCH_FAST(8); }
        else { APAP_LAUNCH_FAST(2); }
#undef APAP_LAUNCH_FAST
    } else if (strips) {
        ProfScope prof(ctx, APAP_PROF_WARP, s);
        const int rows = warp_kernel >= 8 ? 8 : warp_kernel >= 4 ? 4 : 2;  // rows per wave: instantiated for 2, 4, 8
        const dim3 grid((unsigned)((final_w + 255) / 256), (unsigned)((row_count + 4 * rows - 1) / (4 * rows)), batch);
#define APAP_LAUNCH_ROWS(R)                                                                                          \
    if (d_center)                                                                                                    \
        hipLaunchKernelGGL((k_warp_rows<true, R>), grid, dim3(256), 0, s, d_img, img_h, img_w, hinv_pad, mesh_cols,  \
                           lut, final_w, final_h, off_x, off_y, d_out, d_center, center_h, center_w, row_begin,      \
                           row_count, st, mesh_rows, a.status);                                                      \
    else                                                                                                             \
        hipLaunchKernelGGL((k_warp_rows<false, R>), grid, dim3(256), 0, s, d_img, img_h, img_w, hinv_pad, mesh_cols, \
                           lut, final_w, final_h, off_x, off_y, d_out, (const uint8_t *)nullptr, 0, 0, row_begin,    \
                           row_count, st, mesh_rows, a.status)
        if (rows == 4) { APAP_LAUNCH_ROWS(4); }
        else if (rows == 8) { APAP_LAUNCH_ROWS(8); }
        else { APAP_LAUNCH_ROWS(2); }
#undef APAP_LAUNCH_ROWS
    } else {
        ProfScope prof(ctx, APAP_PROF_WARP, s);
        const dim3 grid((unsigned)((threads + 255) / 256), 1, batch);
        if (d_center)
            hipLaunchKernelGGL(k_warp<true>, grid, dim3(256), 0, s, d_img, img_h, img_w, hinv_pad, mesh_cols, lut, final_w,
                               final_h, off_x, off_y, d_out, d_center, center_h, center_w, row_begin, row_count, st, mesh_rows,
                               a.status);
        else
            hipLaunchKernelGGL(k_warp<false>, grid, dim3(256), 0, s, d_img, img_h, img_w, hinv_pad, mesh_cols, lut, final_w,
                               final_h, off_x, off_y, d_out, (const uint8_t *)nullptr, 0, 0, row_begin, row_count, st, mesh_rows,
                               a.status);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "apap_warp_device launch");
    return APAP_OK;
}

// the argument block of the single-pair entry points
WarpArgs single_pair(const uint8_t *d_img, int img_h, int img_w, const uint8_t *d_center, int center_h, int center_w,
                     const void *d_Hfwd, int mesh_rows, int mesh_cols, const double *d_mesh_w, int n_w, const double *d_mesh_h,
                     int n_h, int final_w, int final_h, int off_x, int off_y, uint8_t *d_out, void *d_Hinv_out, void *d_work,
                     size_t work_bytes, int *d_status, void *stream, int row_begin, int row_count) {
    WarpArgs a;
    a.img = d_img; a.img_h = img_h; a.img_w = img_w;
    a.center = d_center; a.center_h = center_h; a.center_w = center_w;
    a.Hfwd = d_Hfwd; a.mesh_rows = mesh_rows; a.mesh_cols = mesh_cols;
    a.mesh_w = d_mesh_w; a.n_w = n_w; a.mesh_h = d_mesh_h; a.n_h = n_h;
    a.final_w = final_w; a.final_h = final_h; a.off_x = off_x; a.off_y = off_y;
    a.out = d_out; a.Hinv_out = d_Hinv_out;
    a.work = d_work; a.work_bytes = work_bytes; a.status = d_status; a.stream = stream;
    a.row_begin = row_begin; a.row_count = row_count;
    return a;
}

}  // namespace

namespace apap {
int warp_phase(apap_ctx *ctx, const uint8_t *d_img, int img_h, int img_w, const uint8_t *d_center, int center_h, int center_w,
               const float *d_Hfwd, int mesh_rows, int mesh_cols, const double *d_mesh_w, int n_w, const double *d_mesh_h,
               int n_h, int final_w, int final_h, int off_x, int off_y, uint8_t *d_out_band, float *d_Hinv_out, void *d_work,
               size_t work_bytes, int *d_status, void *stream, int row_begin, int row_count, int phase, int **d_src_rows) {
    WarpArgs a = single_pair(d_img, img_h, img_w, d_center, center_h, center_w, d_Hfwd, mesh_rows, mesh_cols, d_mesh_w, n_w,
                             d_mesh_h, n_h, final_w, final_h, off_x, off_y, d_out_band, d_Hinv_out, d_work, work_bytes, d_status,
                             stream, row_begin, row_count);
    a.src_rows = d_src_rows;
    if (phase == 0) {       // where the source-row intervals of this workspace live, nothing launched
        if (d_src_rows) {
            const bool fast = n_w <= kMaxEdges && n_h <= kMaxEdges && mesh_rows < 65535 && mesh_cols < 65535;
            *d_src_rows = fast ? warp_work_layout(d_work, mesh_rows, mesh_cols, final_w, final_h).src_rows : nullptr;
        }
        return APAP_OK;
    }
    a.phase = phase;
    return warp_impl<float>(ctx, a);
}
}  // namespace apap

extern "C" {

int apap_warp_device(apap_ctx *ctx, const uint8_t *d_img, int img_h, int img_w, const float *d_Hfwd, int mesh_rows,
                     int mesh_cols, const double *d_mesh_w, int n_w, const double *d_mesh_h,
                     int n_h, int final_w, int final_h, int off_x, int off_y, uint8_t *d_out,
                     float *d_Hinv_out, void *d_work, size_t work_bytes, int *d_status,
                     void *stream) {
    return warp_impl<float>(ctx, single_pair(d_img, img_h, img_w, nullptr, 0, 0, d_Hfwd, mesh_rows, mesh_cols, d_mesh_w, n_w, d_mesh_h,
                                             n_h, final_w, final_h, off_x, off_y, d_out, d_Hinv_out, d_work, work_bytes, d_status,
                                             stream, 0, final_h));
}

int apap_warp_f64_device(apap_ctx *ctx, const uint8_t *d_img, int img_h, int img_w, const double *d_Hfwd, int mesh_rows,
                         int mesh_cols, const double *d_mesh_w, int n_w, const double *d_mesh_h, int n_h, int final_w,
                         int final_h, int off_x, int off_y, uint8_t *d_out, double *d_Hinv_out, void *d_work,
                         size_t work_bytes, int *d_status, void *stream) {
    return warp_impl<double>(ctx, single_pair(d_img, img_h, img_w, nullptr, 0, 0, d_Hfwd, mesh_rows, mesh_cols, d_mesh_w, n_w,
                                              d_mesh_h, n_h, final_w, final_h, off_x, off_y, d_out, d_Hinv_out, d_work, work_bytes,
                                              d_status, stream, 0, final_h));
}

int apap_warp_rows_device(apap_ctx *ctx, const uint8_t *d_img, int img_h, int img_w, const float *d_Hfwd, int mesh_rows,
                          int mesh_cols, const double *d_mesh_w, int n_w, const double *d_mesh_h, int n_h,
                          int final_w, int final_h, int off_x, int off_y, int row_begin, int row_count,
                          uint8_t *d_out_band, void *d_work, size_t work_bytes, int *d_status, void *stream) {
    return warp_impl<float>(ctx, single_pair(d_img, img_h, img_w, nullptr, 0, 0, d_Hfwd, mesh_rows, mesh_cols, d_mesh_w, n_w, d_mesh_h,
                                             n_h, final_w, final_h, off_x, off_y, d_out_band, nullptr, d_work, work_bytes, d_status,
                                             stream, row_begin, row_count));
}

int apap_stitch_device(apap_ctx *ctx, const uint8_t *d_img, int img_h, int img_w, const uint8_t *d_center, int center_h,
                       int center_w, const float *d_Hfwd, int mesh_rows, int mesh_cols,
                       const double *d_mesh_w, int n_w, const double *d_mesh_h, int n_h, int final_w,
                       int final_h, int off_x, int off_y, uint8_t *d_out, float *d_Hinv_out, void *d_work,
                       size_t work_bytes, int *d_status, void *stream) {
    if (!d_center) return apap::fail(APAP_ERR_INVALID_ARG, "apap_stitch_device: null centre image");
    return warp_impl<float>(ctx, single_pair(d_img, img_h, img_w, d_center, center_h, center_w, d_Hfwd, mesh_rows, mesh_cols, d_mesh_w,
                                             n_w, d_mesh_h, n_h, final_w, final_h, off_x, off_y, d_out, d_Hinv_out, d_work, work_bytes,
                                             d_status, stream, 0, final_h));
}

int apap_warp_batch_device(apap_ctx *ctx, const uint8_t *d_imgs, long long img_stride, int img_h, int img_w,
                           const uint8_t *d_centers, long long center_stride, int center_h, int center_w,
                           const float *d_Hfwd, int mesh_rows, int mesh_cols, const double *d_mesh_w, int n_w,
                           const double *d_mesh_h, int n_h, int final_w, int final_h, int off_x, int off_y,
                           int row_begin, int row_count, uint8_t *d_outs, long long out_stride, float *d_Hinv_out,
                           int batch, int phases, void *d_work, size_t work_bytes, int *d_status, void *stream) {
    if (batch < 1 || batch > 65535) return apap::fail(APAP_ERR_INVALID_ARG, "apap_warp_batch_device: batch=%d", batch);
    if (img_stride < 0 || out_stride < 0 || center_stride < 0)
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_warp_batch_device: negative stride");
    if (batch > 1 && (phases & APAP_WARP_GATHER) && out_stride < (long long)row_count * final_w * 3)
        return apap::fail(APAP_ERR_INVALID_ARG, "apap_warp_batch_device: canvases overlap (stride %lld < %lld bytes)", out_stride,
                          (long long)row_count * final_w * 3);
    WarpArgs a = single_pair(d_imgs, img_h, img_w, d_centers, center_h, center_w, d_Hfwd, mesh_rows, mesh_cols, d_mesh_w, n_w,
                             d_mesh_h, n_h, final_w, final_h, off_x, off_y, d_outs, d_Hinv_out, d_work, work_bytes, d_status,
                             stream, row_begin, row_count);
    a.img_stride = img_stride;
    a.center_stride = center_stride;
    a.out_stride = out_stride;
    a.batch = batch;
    a.phase = phases;
    return warp_impl<float>(ctx, a);
}

int apap_warp_coords_device(apap_ctx *ctx, const float *d_Hfwd, int mesh_rows, int mesh_cols, const double *d_mesh_w,
                            int n_w, const double *d_mesh_h, int n_h, int final_w, int final_h,
                            int off_x, int off_y, double *d_coords, void *d_work, size_t work_bytes,
                            int *d_status, void *stream) {
    if (!d_coords) return apap::fail(APAP_ERR_INVALID_ARG, "apap_warp_coords_device: null output");
    hipStream_t s = (hipStream_t)stream;
    WarpWork ww;
    bool fast_tables;
    WarpArgs a = single_pair(nullptr, 0, 0, nullptr, 0, 0, d_Hfwd, mesh_rows, mesh_cols, d_mesh_w, n_w, d_mesh_h, n_h, final_w,
                             final_h, off_x, off_y, nullptr, nullptr, d_work, work_bytes, d_status, stream, 0, final_h);
    a.phase = APAP_WARP_GEOMETRY | APAP_WARP_CELLS;
    const int rc = warp_prologue<float>(ctx, a, &ww, &fast_tables);
    if (rc != APAP_OK) return rc;
    double *hinv_pad = ww.hinv_pad;
    int *lut = ww.lut;
    const size_t total = (size_t)final_w * final_h;
    hipLaunchKernelGGL(k_warp_coords, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, hinv_pad, mesh_cols,
                       lut, final_w, final_h, off_x, off_y, d_coords);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "apap_warp_coords_device launch");
    return APAP_OK;
}

int apap_flatten_device(apap_ctx *ctx, const float *d_H, int cells, double *d_out, int *d_status, void *stream) {
    if (!d_H || !d_out || !d_status) return apap::fail(APAP_ERR_INVALID_ARG, "apap_flatten_device: null device pointer");
    if (cells < 1) return apap::fail(APAP_ERR_INVALID_ARG, "apap_flatten_device: cells=%d", cells);
    hipLaunchKernelGGL(k_flatten, dim3((cells + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_H, cells, d_out,
                       d_status);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "apap_flatten_device launch");
    return APAP_OK;
}

int apap_blend_device(apap_ctx *ctx, const uint8_t *d_a, const uint8_t *d_b, int h, int w, uint8_t *d_out, void *stream) {
    if (!d_a || !d_b || !d_out || h < 1 || w < 1) return apap::fail(APAP_ERR_INVALID_ARG, "apap_blend_device: bad argument");
    const size_t pixels = (size_t)h * w;
    const int blocks = (int)((pixels + 255) / 256 < 16384 ? (pixels + 255) / 256 : 16384);
    hipLaunchKernelGGL(k_blend, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_a, d_b, pixels, d_out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "apap_blend_device launch");
    return APAP_OK;
}

}  // extern "C"
