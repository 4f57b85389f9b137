"""Algorithmic (compulsory) HBM bytes of the path, SURVEY.md section 8(d):
every array streamed once per logical op, fp64 values + int32 indices.  These
are the figures ``bench.py`` divides by measured time; a fused kernel that
moves fewer bytes simply scores a higher fraction - the definitions are never
adjusted downward."""

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_MEASURED_GBS = 6290.0        # float4 copy, same guide


def b_spmv(nr, nc, nnz):
    return 12 * nnz + 4 * (nr + 1) + 8 * nr + 8 * nc


def b_axpy(n):
    return 24 * n


def b_dot(n):
    return 16 * n


def b_copy(n):
    return 16 * n


def b_bc(n_bc):
    return 12 * n_bc


def b_cg(n, nnz):
    """one Jacobi-PCG iteration (unfused textbook count)"""
    return 12 * nnz + 148 * n + 4


def b_cheb(n, nnz):
    """one Chebyshev-Jacobi iteration"""
    return 12 * nnz + 92 * n + 4


def b_inner(ksp, n, nnz, its):
    if ksp == "cg":
        return its * b_cg(n, nnz)
    if ksp in ("chebyshev", "richardson"):
        return its * b_cheb(n, nnz)
    return b_copy(n)             # preonly + jacobi


def b_pcd(n_p, nnz_ap, nnz_mp, nnz_kp, n_bc, k_a, k_m, ksp_a="cg",
          ksp_m="chebyshev"):
    """BRM1 apply: copy + bc + Ap solve + Kp SpMV + axpy + Mp solve + scale"""
    return (b_copy(n_p) + b_bc(n_bc) + b_inner(ksp_a, n_p, nnz_ap, k_a)
            + b_spmv(n_p, n_p, nnz_kp) + b_axpy(n_p)
            + b_inner(ksp_m, n_p, nnz_mp, k_m) + b_copy(n_p))


def b_fieldsplit(n_u, n_p, nnz_a00, nnz_a01, pcd_bytes, k_f,
                 ksp_f="chebyshev"):
    return (pcd_bytes + b_spmv(n_u, n_p, nnz_a01) + b_axpy(n_u)
            + b_inner(ksp_f, n_u, nnz_a00, k_f) + 16 * (n_u + n_p))


def b_vcycle(mg_data, n_fine, nnz_fine):
    """One V(nu, nu) cycle (pcd_apply.hip: mg_vcycle / mg_smooth): per level
    l >= 1 a Jacobi start (24 n), nu-1 + nu fused Chebyshev steps, the
    residual SpMV, restriction and prolongation-add SpMVs; level 0 is one
    dense SpMV; one final copy on the finest level."""
    ops, chain, nu = mg_data["ops"], mg_data["chain"], mg_data["nu"]
    nu2 = mg_data.get("nu_post", nu)
    L = len(ops)
    total = b_spmv(ops[0].shape[0], ops[0].shape[0], mg_data["C"].nnz)
    for l in range(1, L):
        n = n_fine if l == L - 1 else ops[l].shape[0]
        nnz = nnz_fine if l == L - 1 else ops[l].nnz
        P = chain[l]
        total += 24 * n + (nu + nu2 - 1) * b_cheb(n, nnz)
        total += b_spmv(n, n, nnz) + 8 * n
        total += b_spmv(P.shape[1], P.shape[0], P.nnz)
        total += b_spmv(P.shape[0], P.shape[1], P.nnz) + 8 * n
    return total + b_copy(n_fine)
