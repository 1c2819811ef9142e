//! Rust side of the drop-in boundary: the FFI declarations of `include/summa_gpu.h` and safe
//! wrappers with the exact signatures of the two `halo2_proofs::arithmetic` functions the
//! prover's hot path goes through (`best_multiexp`, `best_fft`), plus the `EvaluationDomain` /
//! `ParamsKZG` conveniences.  UNBUILT in this repository (no cargo/rustc in the image); the same
//! ABI is exercised through `circuits_halo2_amd/ffi.py`.
//!
//! Layout contract: `halo2curves::bn256::Fr` is 4 x u64 little-endian limbs in Montgomery form
//! (32 bytes), `G1Affine` is `x || y` of two such `Fq` (64 bytes, identity = all zero); the
//! assertions below make a layout change a compile error instead of a wrong proof.
use halo2curves::bn256::{Fr, G1Affine, G1};
use std::os::raw::{c_char, c_int, c_void};

#[allow(non_camel_case_types)]
type size_t = usize;

extern "C" {
    pub fn sg_init(device: c_int) -> c_int;
    pub fn sg_shutdown();
    pub fn sg_last_error() -> *const c_char;
    pub fn sg_device_count() -> c_int;
    pub fn sg_device() -> c_int;
    pub fn sg_bind_thread() -> c_int;
    // host memory that stays in place (an SRS, an arena of columns) can be page-locked once for the host-pointer entry points
    pub fn sg_host_register(host: *mut c_void, bytes: size_t) -> c_int;
    pub fn sg_host_unregister(host: *mut c_void) -> c_int;
    pub fn sg_stream_wait(stream: *mut c_void) -> c_int;
    pub fn sg_set_param(name: *const c_char, value: c_int) -> c_int;
    pub fn sg_msm_g1(scalars: *const u8, bases: *const u8, n: size_t, out_affine: *mut u8) -> c_int;
    pub fn sg_msm_g1_batch(
        scalars: *const *const u8,
        bases: *const *const u8,
        n: *const size_t,
        count: size_t,
        out_affine: *mut u8,
    ) -> c_int;
    pub fn sg_srs_upload(k: u32, g: *const u8, g_lagrange: *const u8, handle_out: *mut u64) -> c_int;
    pub fn sg_srs_free(handle: u64) -> c_int;
    pub fn sg_srs_precompute(handle: u64, basis: c_int, window_bits: u32) -> c_int;
    pub fn sg_commit(handle: u64, basis: c_int, scalars: *const u8, n: size_t, out_affine: *mut u8) -> c_int;
    pub fn sg_ntt_fr(a: *mut u8, omega: *const u8, log_n: u32) -> c_int;
    pub fn sg_intt_fr(a: *mut u8, omega_inv: *const u8, divisor: *const u8, log_n: u32) -> c_int;
    pub fn sg_coeff_to_extended(coeffs: *const u8, k: u32, ext_k: u32, out: *mut u8) -> c_int;
    pub fn sg_extended_to_coeff(ext: *mut u8, k: u32, ext_k: u32) -> c_int;
    pub fn sg_divide_by_vanishing_poly(ext: *mut u8, k: u32, ext_k: u32) -> c_int;
    pub fn sg_fr_eval_poly(coeffs: *const u8, n: size_t, x: *const u8, out: *mut u8) -> c_int;
    pub fn sg_kzg_setup(k: u32, tau: *const u8, g: *mut u8, g_lagrange: *mut u8) -> c_int;
    pub fn sg_g1_to_lagrange(g: *const u8, k: u32, g_lagrange: *mut u8) -> c_int;
    // device-resident variants (`*_dev`) take HIP pointers; a Rust prover that keeps its
    // polynomials in HBM binds them the same way (see include/summa_gpu.h)
    pub fn sg_msm_g1_dev(s: *const c_void, b: *const c_void, n: size_t, stream: *mut c_void, out: *mut u8) -> c_int;
    pub fn sg_commit_batch_dev(
        handle: u64,
        basis: c_int,
        d_scalars: *const *const c_void,
        count: size_t,
        n: size_t,
        stream: *mut c_void,
        out_affine: *mut u8,
    ) -> c_int;
    pub fn sg_commit_batch_mixed_dev(
        handle: u64,
        basis: *const c_int,
        d_scalars: *const *const c_void,
        count: size_t,
        n: size_t,
        stream: *mut c_void,
        out_affine: *mut u8,
    ) -> c_int;
    pub fn sg_fr_eval_poly_batch_dev(
        d_polys: *const *const c_void,
        n: size_t,
        points: *const u8,
        m: u32,
        stream: *mut c_void,
        out: *mut u8,
    ) -> c_int;
    pub fn sg_fr_kate_division_dev(d_a: *const c_void, n: size_t, b: *const u8, d_q: *mut c_void, rem: *mut u8, stream: *mut c_void) -> c_int;
    pub fn sg_g2_generator_mul(scalar: *const u8, out128: *mut u8) -> c_int;
    // blinding factors / the random polynomial (OsRng upstream): a 32-byte OS-random key expanded by ChaCha20 on the device
    pub fn sg_fr_random_dev(key: *const u8, stream_id: u64, d_out: *mut c_void, n: size_t, stream: *mut c_void) -> c_int;
    // permutation::prover::commit over every chunk + lookup::prover::commit_product, one batched call per proof
    pub fn sg_grand_products_dev(
        d_values: *const *const c_void,
        d_sigma: *const *const c_void,
        chunk_cols: *const u32,
        n_chunks: u32,
        d_lookup_cols: *const *const c_void,
        n_lookups: u32,
        beta: *const u8,
        gamma: *const u8,
        k: u32,
        usable_rows: size_t,
        d_z: *const *mut c_void,
        stream: *mut c_void,
    ) -> c_int;
    // lookup::prover::permute_expression_pair for range tables (returns -5 when the table is not small integers)
    pub fn sg_lookup_permute_small_dev(
        d_input: *const c_void,
        d_table: *const c_void,
        rows: size_t,
        d_permuted_input: *mut c_void,
        d_permuted_table: *mut c_void,
        stream: *mut c_void,
    ) -> c_int;
}

/// `sg_graph` of include/summa_gpu.h: the plain-struct image of halo2's `GraphEvaluator` (what `pk.ev` holds)
#[repr(C)]
pub struct SgGraph {
    _opaque: [u8; 0], // laid out by include/summa_gpu.h; built by the code that walks `GraphEvaluator` (not part of this shim)
}

// The compiled-host prover and verifier for `MstInclusionCircuit` (include/summa_prover.h): what
// `zk_prover/src/circuits/utils.rs` calls in place of `create_proof` / `verify_proof`
extern "C" {
    pub fn sp_key_create(
        k: u32,
        srs_handle: u64,
        d_fixed_lagrange: *const *const c_void,
        d_sigma_lagrange: *const *const c_void,
        vk_digest_be: *const u8,
        gates: *const SgGraph,
        lookup_input: *const SgGraph,
        gate_challenge_exponents: *const u32,
        gate_challenge_counts: *const u32,
        n_gate_challenges: u32,
        stream: *mut c_void,
        key_out: *mut u64,
    ) -> c_int;
    pub fn sp_key_destroy(key: u64) -> c_int;
    pub fn sp_create_proof(
        key: u64,
        d_advice: *const *mut c_void,
        instances: *const u8,
        n_instances: u32,
        transcript: c_int,
        sanity_checks: c_int,
        stream: *mut c_void,
        proof_out: *mut u8,
        proof_cap: size_t,
        proof_len: *mut size_t,
    ) -> c_int;
    pub fn sp_last_error() -> *const c_char;
    pub fn sp_verify_proof(
        k: u32,
        n_currencies: u32,
        vk_digest_be: *const u8,
        fixed_comms: *const u8,
        permutation_comms: *const u8,
        g2: *const u8,
        s_g2: *const u8,
        proof: *const u8,
        proof_len: size_t,
        instances: *const u8,
        n_instances: u32,
        transcript: c_int,
        accepted: *mut c_int,
    ) -> c_int;
    pub fn sp_verify_last_error() -> *const c_char;
    // a setup that arrives in device memory (RCCL broadcast), proofs in flight from several threads, memory hygiene
    pub fn sg_srs_upload_dev(k: u32, d_g: *const c_void, d_g_lagrange: *const c_void, stream: *mut c_void, handle_out: *mut u64) -> c_int;
    pub fn sg_srs_copy_dev(handle: u64, d_g_out: *mut c_void, d_g_lagrange_out: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn sg_commit_combine_begin() -> c_int;
    pub fn sg_commit_combine_end() -> c_int;
    pub fn sg_commit_combine_stats(jobs: *mut u64, requests: *mut u64) -> c_int;
    pub fn sg_collect_retired() -> c_int;
    pub fn sg_set_param(name: *const c_char, value: c_int) -> c_int;
    pub fn sg_get_param(name: *const c_char, value: *mut c_int) -> c_int;
    // round 5: a proof without copy launches -- small results land in memory the caller names (page-locked, mapped into the device)
    pub fn sg_fr_flag_noncanonical_dev(d_cols: *const *const c_void, m: u32, n: size_t, d_flag: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn sg_lookup_permute_small_async_dev(
        d_input: *const c_void,
        d_table: *const c_void,
        rows: size_t,
        d_permuted_input: *mut c_void,
        d_permuted_table: *mut c_void,
        d_status: *mut c_void,
        stream: *mut c_void,
    ) -> c_int;
    pub fn sg_fr_kate_division_rem_dev(d_a: *const c_void, n: size_t, b: *const u8, d_q: *mut c_void, d_remainder: *mut c_void, stream: *mut c_void) -> c_int;
    // ... and the rotation sets' combinations of the multi-open in one launch
    pub fn sg_fr_lincomb_sets_dev(
        d_polys: *const *const c_void,
        coeffs: *const u8,
        set_sizes: *const u32,
        n_sets: u32,
        n: size_t,
        lows: *const u8,
        n_lows: *const u32,
        d_outs: *const *mut c_void,
        stream: *mut c_void,
    ) -> c_int;
    // revision of include/summa_gpu.h the library was built from (SG_ABI_VERSION); this file is written against 3
    pub fn sg_abi_version() -> c_int;
}

const _: () = assert!(std::mem::size_of::<Fr>() == 32);
const _: () = assert!(std::mem::size_of::<G1Affine>() == 64);

#[derive(Debug)]
pub struct GpuError(pub c_int, pub String);

fn check(rc: c_int) -> Result<(), GpuError> {
    if rc == 0 {
        return Ok(());
    }
    let msg = unsafe { std::ffi::CStr::from_ptr(sg_last_error()) }.to_string_lossy().into_owned();
    Err(GpuError(rc, msg))
}

fn affine_from_bytes(b: [u8; 64]) -> G1Affine {
    // 64 zero bytes = identity (halo2curves' own encoding of the point at infinity);
    // otherwise (x, y) Montgomery limbs, which is G1Affine's in-memory representation
    unsafe { std::mem::transmute::<[u8; 64], G1Affine>(b) }
}

/// `halo2_proofs::arithmetic::best_multiexp` for `C = bn256::G1Affine`.
pub fn best_multiexp(coeffs: &[Fr], bases: &[G1Affine]) -> Result<G1, GpuError> {
    assert_eq!(coeffs.len(), bases.len()); // upstream's own assertion
    let mut out = [0u8; 64];
    check(unsafe { sg_msm_g1(coeffs.as_ptr() as *const u8, bases.as_ptr() as *const u8, coeffs.len(), out.as_mut_ptr()) })?;
    Ok(G1::from(affine_from_bytes(out)))
}

/// The commitments of one prover phase in one call (advice columns, quotient pieces, ...).
pub fn best_multiexp_batch(pairs: &[(&[Fr], &[G1Affine])]) -> Result<Vec<G1>, GpuError> {
    let s: Vec<*const u8> = pairs.iter().map(|(c, _)| c.as_ptr() as *const u8).collect();
    let b: Vec<*const u8> = pairs.iter().map(|(_, g)| g.as_ptr() as *const u8).collect();
    let n: Vec<usize> = pairs.iter().map(|(c, g)| { assert_eq!(c.len(), g.len()); c.len() }).collect();
    let mut out = vec![0u8; 64 * pairs.len()];
    check(unsafe { sg_msm_g1_batch(s.as_ptr(), b.as_ptr(), n.as_ptr(), pairs.len(), out.as_mut_ptr()) })?;
    Ok(out.chunks_exact(64).map(|c| G1::from(affine_from_bytes(c.try_into().unwrap()))).collect())
}

/// `halo2_proofs::arithmetic::best_fft` for `Scalar = G = bn256::Fr` (in place, natural order).
pub fn best_fft(a: &mut [Fr], omega: Fr, log_n: u32) -> Result<(), GpuError> {
    assert_eq!(a.len(), 1usize << log_n);
    check(unsafe { sg_ntt_fr(a.as_mut_ptr() as *mut u8, &omega as *const Fr as *const u8, log_n) })
}

/// `EvaluationDomain::ifft(a, omega_inv, log_n, divisor)`.
pub fn ifft(a: &mut [Fr], omega_inv: Fr, log_n: u32, divisor: Fr) -> Result<(), GpuError> {
    assert_eq!(a.len(), 1usize << log_n);
    check(unsafe {
        sg_intt_fr(a.as_mut_ptr() as *mut u8, &omega_inv as *const Fr as *const u8, &divisor as *const Fr as *const u8, log_n)
    })
}

/// `EvaluationDomain::coeff_to_extended` (zeta-coset, zero padding and the NTT fused).
pub fn coeff_to_extended(coeffs: &[Fr], k: u32, extended_k: u32) -> Result<Vec<Fr>, GpuError> {
    assert_eq!(coeffs.len(), 1usize << k);
    let mut out = vec![Fr::zero(); 1usize << extended_k];
    check(unsafe { sg_coeff_to_extended(coeffs.as_ptr() as *const u8, k, extended_k, out.as_mut_ptr() as *mut u8) })?;
    Ok(out)
}

/// `EvaluationDomain::extended_to_coeff`; the caller truncates to `n * quotient_poly_degree`.
pub fn extended_to_coeff(ext: &mut [Fr], k: u32, extended_k: u32) -> Result<(), GpuError> {
    assert_eq!(ext.len(), 1usize << extended_k);
    check(unsafe { sg_extended_to_coeff(ext.as_mut_ptr() as *mut u8, k, extended_k) })
}

/// SRS kept resident in HBM: upload once in `ParamsKZG::read/setup`, then `commit`/`commit_lagrange`.
pub struct SrsHandle(u64);
impl SrsHandle {
    pub fn upload(k: u32, g: &[G1Affine], g_lagrange: &[G1Affine]) -> Result<Self, GpuError> {
        assert!(g.len() == 1usize << k && g_lagrange.len() == g.len());
        let mut h = 0u64;
        check(unsafe { sg_srs_upload(k, g.as_ptr() as *const u8, g_lagrange.as_ptr() as *const u8, &mut h) })?;
        Ok(SrsHandle(h))
    }
    /// Build the fixed-base window tables once (both bases); every later commit takes the fixed-base path.
    pub fn precompute(&self) -> Result<(), GpuError> {
        check(unsafe { sg_srs_precompute(self.0, 0, 0) })?;
        check(unsafe { sg_srs_precompute(self.0, 1, 0) })
    }
    /// basis: false = `ParamsKZG::commit` (monomial g), true = `commit_lagrange`
    pub fn commit(&self, lagrange: bool, poly: &[Fr]) -> Result<G1, GpuError> {
        let mut out = [0u8; 64];
        check(unsafe { sg_commit(self.0, lagrange as c_int, poly.as_ptr() as *const u8, poly.len(), out.as_mut_ptr()) })?;
        Ok(G1::from(affine_from_bytes(out)))
    }
}
impl Drop for SrsHandle {
    fn drop(&mut self) {
        unsafe { sg_srs_free(self.0) };
    }
}
