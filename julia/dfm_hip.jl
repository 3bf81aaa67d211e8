# julia/dfm_hip.jl -- the thin `ccall` shim between QuantEcon/dynamic_factor_models' Julia code and
# libdfmhip.so (include/dfm_hip.h).  It fills the dispatch slot the reference declares but leaves empty:
# `struct Parametric <: EstimationMethod end` (dfm_functions.ipynb:21-23); the reference implements only
# `estimate!(m, ::NonParametric)` (dfm_functions.ipynb:530-543).
#
# Usage (in the notebook's working directory, after the reference's own includes so that DFMModel,
# Parametric, standardize_data and drop_missing_col exist -- Stock_Watson.ipynb:39-41):
#     include("readin_functions.jl"); @nbinclude("dfm_functions.ipynb")
#     include("julia/dfm_hip.jl")                      # this file; needs ENV["DFMHIP_LIB"] or the default path
#     estimate!(dfmm, Parametric(); max_em_iter = 50)  # instead of estimate!(dfmm, NonParametric())
# Nothing else in the notebook changes: `Stock_Watson.ipynb` never passes `Parametric()` itself, so it keeps
# running unchanged on the pure-Julia path.
#
# NOT EXECUTED IN THE BUILD IMAGE (no Julia there): kept deliberately thin -- every numerical step is a
# single ccall; the same sequence of calls is exercised by dynamic_factor_models_amd/api.py (tests/
# test_gpu_api.py), which mirrors this file line by line.

module DFMHip

const LIB = get(ENV, "DFMHIP_LIB", joinpath(@__DIR__, "..", "dynamic_factor_models_amd", "lib", "libdfmhip.so"))
const DFM_F_MAY_HAVE_MISSING = Cuint(1)

mutable struct Handle
    ptr::Ptr{Cvoid}
end

function check(h::Ptr{Cvoid}, rc::Cint)
    rc == 0 && return nothing
    msg = h == C_NULL ? "no handle" : unsafe_string(ccall((:dfm_last_error, LIB), Cstring, (Ptr{Cvoid},), h))
    error("libdfmhip: status $rc: $msg")            # same style as the reference's error("...") (dfm_functions.ipynb:124-126)
end

function create(device::Integer = 0)
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:dfm_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Cint, Ptr{Cvoid}), ref, device, C_NULL)
    rc == 0 || error("libdfmhip: dfm_create failed with status $rc (no HIP device?)")
    h = Handle(ref[])
    finalizer(x -> (x.ptr != C_NULL && ccall((:dfm_destroy, LIB), Cint, (Ptr{Cvoid},), x.ptr); x.ptr = C_NULL), h)
    return h
end

# The C side wants panel[b][t][i] with i fastest.  A Julia Array{Float64,3} of size (N, T, B) has exactly
# that memory order, so the shim permutes the reference's T x ns matrix once: permutedims(x, (2, 1)).
to_c_panel(z::AbstractMatrix{Float64}) = reshape(permutedims(z, (2, 1)), size(z, 2), size(z, 1), 1)
nan_for_missing(x) = Float64[ismissing(v) ? NaN : Float64(v) for v in x]       # Union{Missing,Float64} is not C layout

"PCA + OLS start of EM on a balanced standardised T x N panel (dfm_pca_init_batch)."
function pca_init(h::Handle, xbal::Matrix{Float64}, r::Integer)
    T, N = size(xbal)
    panel = to_c_panel(xbal)
    Lam = Array{Float64}(undef, r, N, 1); R = Array{Float64}(undef, N, 1)
    A = Array{Float64}(undef, r, r, 1); Q = similar(A); P0 = similar(A); mu0 = Array{Float64}(undef, r, 1)
    F = Array{Float64}(undef, r, T, 1)
    GC.@preserve panel Lam R A Q mu0 P0 F begin
        rc = ccall((:dfm_pca_init_batch, LIB), Cint,
                   (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                   h.ptr, 1, T, N, r, panel, Lam, R, A, Q, mu0, P0, F)
        check(h.ptr, rc)
    end
    # C row-major [i][k] == Julia column-major (k, i): transpose back to the reference's ns x r / T x r
    return (Lam = permutedims(Lam[:, :, 1]), R = R[:, 1], A = permutedims(A[:, :, 1]), Q = permutedims(Q[:, :, 1]),
            mu0 = mu0[:, 1], P0 = permutedims(P0[:, :, 1]), F = permutedims(F[:, :, 1]))
end

"max_iter EM iterations from the given start (dfm_em_batch); z is T x N with NaN for missing."
function em(h::Handle, z::Matrix{Float64}, p; max_iter::Integer = 50, tol::Real = 1e-6)
    T, N = size(z); r = size(p.Lam, 2)
    panel = to_c_panel(z)
    Lam = reshape(permutedims(p.Lam), r, N, 1); R = reshape(copy(p.R), N, 1)
    A = reshape(permutedims(p.A), r, r, 1); Q = reshape(permutedims(p.Q), r, r, 1)
    mu0 = reshape(copy(p.mu0), r, 1); P0 = reshape(permutedims(p.P0), r, r, 1)
    path = Array{Float64}(undef, max_iter, 1); iters = Array{Cint}(undef, 1)
    f = Array{Float64}(undef, r, T, 1); np = div(r * (r + 1), 2); P = Array{Float64}(undef, np, T, 1)
    flags = any(isnan, z) ? DFM_F_MAY_HAVE_MISSING : Cuint(0)
    GC.@preserve panel Lam R A Q mu0 P0 path iters f P begin
        rc = ccall((:dfm_em_batch, LIB), Cint,
                   (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Cdouble, Ptr{Float64}, Ptr{Cint},
                    Ptr{Float64}, Ptr{Float64}, Cuint),
                   h.ptr, 1, T, N, r, panel, Lam, R, A, Q, mu0, P0, max_iter, tol, path, iters, f, P, flags)
        check(h.ptr, rc)
    end
    k = Int(iters[1])
    return (Lam = permutedims(Lam[:, :, 1]), R = R[:, 1], A = permutedims(A[:, :, 1]), Q = permutedims(Q[:, :, 1]),
            mu0 = mu0[:, 1], P0 = permutedims(P0[:, :, 1]), loglik = path[1:k, 1], iters = k,
            factor = permutedims(f[:, :, 1]))
end

end # module

# ---------------------------------------------------------------------------------------------------------
# The new method.  Same mutate-in-place convention as the reference's estimate! (dfm_functions.ipynb:530-543);
# additionally returns the per-iteration log-likelihood vector.
function estimate!(m::DFMModel, ::Parametric; max_em_iter::Integer = 50, tol_em::Real = 1e-6,
                   device::Integer = 0, handle = nothing)
    m.nfac_o == 0 || error("observed factors are not supported on the parametric path")
    r = m.nfac_u
    incl = m.inclcode .== 1
    xdata = m.data[m.initperiod:m.lastperiod, incl]                       # dfm_functions.ipynb:335-336
    xstd, xsd = standardize_data(xdata)                                   # :339
    m.fes.tss = sum(skipmissing(xstd .^ 2))                               # :342
    m.fes.nobs = count(.!ismissing.(xstd))                                # :343
    z = reshape(DFMHip.nan_for_missing(xstd), size(xstd))
    xbal, balmask = drop_missing_col(xstd)                                # :345
    balmask = vec(balmask)
    h = handle === nothing ? DFMHip.create(device) : handle
    p0 = DFMHip.pca_init(h, Float64.(xbal), r)                            # pca_score (:179-183) + OLS start, on the GPU
    N = size(z, 2)
    Lam = Matrix{Float64}(undef, N, r); R = Vector{Float64}(undef, N)
    Lam[balmask, :] = p0.Lam; R[balmask] = p0.R
    for i in findall(.!balmask)                                           # series with gaps: complete-case OLS (:242-252)
        b, e, rows = ols_skipmissing(xstd[:, i], p0.F, Balanced())
        Lam[i, :] = b; R[i] = sum(abs2, e) / count(rows)
    end
    fit = DFMHip.em(h, z, (Lam = Lam, R = R, A = p0.A, Q = p0.Q, mu0 = p0.mu0, P0 = p0.P0);
                    max_iter = max_em_iter, tol = tol_em)
    m.factor[m.initperiod:m.lastperiod, :] = fit.factor                   # in place: aliases factor_var_model.y (:80, :371)
    cols = findall(incl)
    m.lambda[cols, :] = fit.Lam .* vec(xsd)
    m.uar_ser[cols] = sqrt.(fit.R) .* vec(xsd)
    m.uar_coef[cols, :] .= 0.0
    common = fit.factor * fit.Lam'
    e = [isnan(z[t, i]) ? 0.0 : z[t, i] - common[t, i] for t in 1:size(z, 1), i in 1:N]
    m.fes.ssr = sum(abs2, e)                                              # :366
    var = m.factor_var_model                                              # fill_matrices! (:477-492), VAR(1) block
    fill!(var.M, 0.0); fill!(var.Q, 0.0); fill!(var.G, 0.0)
    var.M[1:r, 1:r] = fit.A
    var.nlag > 1 && (var.M[r+1:end, 1:end-r] = Matrix(1.0I, r * (var.nlag - 1), r * (var.nlag - 1)))
    var.Q[1:r, 1:r] = Matrix(1.0I, r, r)
    var.seps[:, :] = fit.Q
    var.G[1:r, 1:r] = cholesky(Symmetric(fit.Q)).L
    return fit.loglik
end
