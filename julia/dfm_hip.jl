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
# test_gpu_api.py), which mirrors this file step by step, and tests/test_julia_shim_cpu.py parses every ccall below
# and checks symbol, return type, argument count and argument types against include/dfm_hip.h.

module DFMHip

const LIB = get(ENV, "DFMHIP_LIB", joinpath(@__DIR__, "..", "dynamic_factor_models_amd", "lib", "libdfmhip.so"))
const DFM_F_MAY_HAVE_MISSING = Cuint(1)
const DFM_F_SINGULAR_Q = Cuint(2)
const DFM_E_NUMERIC = Cint(-5)

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

# One handle (context + device workspace) per device for the whole session: the notebook calls the estimator ~200 times
# per table (Stock_Watson.ipynb:516, 591, 643, 893) and a handle per call would allocate and free the workspace each time.
const HANDLES = Dict{Int,Handle}()
function handle(device::Integer = 0)
    h = get(HANDLES, Int(device), nothing)
    (h === nothing || h.ptr == C_NULL) && (h = HANDLES[Int(device)] = create(device))
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
    start = (copy(Lam), copy(R), copy(A), copy(Q), copy(mu0), copy(P0))
    fallback = false
    GC.@preserve panel Lam R A Q mu0 P0 path iters f P begin
        rc = ccall((:dfm_em_batch, LIB), Cint,
                   (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Cdouble, Ptr{Float64}, Ptr{Cint},
                    Ptr{Float64}, Ptr{Float64}, Cuint),
                   h.ptr, 1, T, N, r, panel, Lam, R, A, Q, mu0, P0, max_iter, tol, path, iters, f, P, flags)
        fallback = rc == DFM_E_NUMERIC
        if fallback
            # the information-form recursion inverts Q; a PCA start on fewer than 2r + 1 periods has a rank-deficient VAR
            # residual covariance: run the covariance-form recursion instead (as api.estimate does)
            Lam, R, A, Q, mu0, P0 = start
            rc = ccall((:dfm_em_batch, LIB), Cint,
                       (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                        Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Cdouble, Ptr{Float64}, Ptr{Cint},
                        Ptr{Float64}, Ptr{Float64}, Cuint),
                       h.ptr, 1, T, N, r, panel, Lam, R, A, Q, mu0, P0, max_iter, tol, path, iters, f, P,
                       flags | DFM_F_SINGULAR_Q)
        end
        check(h.ptr, rc)
    end
    k = Int(iters[1])
    return (Lam = permutedims(Lam[:, :, 1]), R = R[:, 1], A = permutedims(A[:, :, 1]), Q = permutedims(Q[:, :, 1]),
            mu0 = mu0[:, 1], P0 = permutedims(P0[:, :, 1]), loglik = path[1:k, 1], iters = k,
            factor = permutedims(f[:, :, 1]), singular_q = fallback)
end

"EM with OBSERVED factors as known regressors (dfm_em_obs_batch; include/dfm_hip.h): z is T x N (NaN = missing), G the
T x r_o observed factors (no gaps), p.Lam N x (r_o + r_u) with the observed-factor loadings FIRST (the reference's column
order, dfm_functions.ipynb:364), p.A / p.Q / p.P0 r_u x r_u, p.mu0 r_u."
function em_obs(h::Handle, z::Matrix{Float64}, G::Matrix{Float64}, p; max_iter::Integer = 50, tol::Real = 1e-6)
    T, N = size(z); ro = size(G, 2); re = size(p.Lam, 2); ru = re - ro
    panel = to_c_panel(z); Gc = reshape(permutedims(G, (2, 1)), ro, T, 1)
    Lam = reshape(permutedims(p.Lam), re, N, 1); R = reshape(copy(p.R), N, 1)
    A = reshape(permutedims(p.A), ru, ru, 1); Q = reshape(permutedims(p.Q), ru, ru, 1)
    mu0 = reshape(copy(p.mu0), ru, 1); P0 = reshape(permutedims(p.P0), ru, ru, 1)
    path = Array{Float64}(undef, max_iter, 1); iters = Array{Cint}(undef, 1)
    f = Array{Float64}(undef, ru, T, 1); np = div(ru * (ru + 1), 2); P = Array{Float64}(undef, np, T, 1)
    flags = any(isnan, z) ? DFM_F_MAY_HAVE_MISSING : Cuint(0)
    GC.@preserve panel Gc Lam R A Q mu0 P0 path iters f P begin
        rc = ccall((:dfm_em_obs_batch, LIB), Cint,
                   (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Cdouble, Ptr{Float64}, Ptr{Cint}, Ptr{Float64}, Ptr{Float64}, Cuint),
                   h.ptr, 1, T, N, ru, ro, panel, Gc, Lam, R, A, Q, mu0, P0, max_iter, tol, path, iters, f, P, flags)
        check(h.ptr, rc)
    end
    k = Int(iters[1])
    return (Lam = permutedims(Lam[:, :, 1]), R = R[:, 1], A = permutedims(A[:, :, 1]), Q = permutedims(Q[:, :, 1]),
            mu0 = mu0[:, 1], P0 = permutedims(P0[:, :, 1]), loglik = path[1:k, 1], iters = k, factor = permutedims(f[:, :, 1]))
end

# C [b][i][k] <-> Julia: a vector of B matrices (rows x cols) -> Array (cols, rows, B), and back
pack3(ms::Vector{Matrix{Float64}}) = cat([permutedims(m) for m in ms]...; dims = 3)
unpack3(a::Array{Float64,3}) = [permutedims(a[:, :, b]) for b in 1:size(a, 3)]

"B smoother passes in ONE call (dfm_ks_pass_batch; with ngpu > 1: dfm_ks_pass_batch_multi, replicates split over the
GPUs): panels[b] is T x N (NaN = missing), params[b] = (Lam N x r, R, A, Q, mu0, P0).  Returns per-replicate smoothed
factors (T x r), packed covariances (T x r(r+1)/2) and log-likelihoods."
function ks_pass_batch(h::Handle, panels::Vector{Matrix{Float64}}, params::Vector; ngpu::Integer = 1)
    B = length(panels); T, N = size(panels[1]); r = size(params[1].Lam, 2)
    panel = cat([permutedims(z, (2, 1)) for z in panels]...; dims = 3)           # (N, T, B) == C [b][t][i]
    Lam = pack3([p.Lam for p in params]); R = hcat([p.R for p in params]...)
    A = pack3([p.A for p in params]); Q = pack3([p.Q for p in params]); P0 = pack3([p.P0 for p in params])
    mu0 = hcat([p.mu0 for p in params]...)
    np = div(r * (r + 1), 2)
    f = Array{Float64}(undef, r, T, B); P = Array{Float64}(undef, np, T, B); ll = Array{Float64}(undef, B)
    flags = any(isnan, panel) ? DFM_F_MAY_HAVE_MISSING : Cuint(0)
    GC.@preserve panel Lam R A Q mu0 P0 f P ll begin
        if ngpu <= 1
            rc = ccall((:dfm_ks_pass_batch, LIB), Cint,
                       (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                        Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cuint),
                       h.ptr, B, T, N, r, panel, Lam, R, A, Q, mu0, P0, f, P, ll, flags)
            check(h.ptr, rc)
        else
            err = zeros(UInt8, 700)
            rc = ccall((:dfm_ks_pass_batch_multi, LIB), Cint,
                       (Cint, Ptr{Cint}, Cint, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                        Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cuint,
                        Ptr{UInt8}, Cint),
                       ngpu, C_NULL, B, T, N, r, panel, Lam, R, A, Q, mu0, P0, f, P, ll, flags, err, length(err))
            rc == 0 || error("libdfmhip: status $rc: $(unsafe_string(pointer(err)))")
        end
    end
    return (factor = unpack3(f), P = unpack3(P), loglik = ll)
end

# ---- the library's multi-GPU object (dfm_multi, csrc/multi.hip): per-GPU handles / streams / workspaces and ONE RCCL
# communicator, created once and cached per GPU count; a job's replicates stay resident on the GPUs between calls -------
mutable struct Multi
    ptr::Ptr{Cvoid}
    shape::NTuple{4,Int}      # (B, T, N, r) of the resident job
    max_iter::Int
end
const DFM_MULTI_F_FORCE_COMM = Cuint(1)
const MULTI_LAM, MULTI_R, MULTI_A, MULTI_Q, MULTI_MU0, MULTI_P0, MULTI_F_SMOOTH, MULTI_P_SMOOTH, MULTI_LOGLIK,
      MULTI_LOGLIK_PATH, MULTI_ITERS, MULTI_PANEL = Cint.(0:11)

function mcheck(m::Multi, rc::Cint)
    rc == 0 && return nothing
    error("libdfmhip: status $rc: " * unsafe_string(ccall((:dfm_multi_last_error, LIB), Cstring, (Ptr{Cvoid},), m.ptr)))
end

function multi_create(ngpu::Integer = 1; device_ids = nothing, force_comm::Bool = false)
    ref = Ref{Ptr{Cvoid}}(C_NULL); err = zeros(UInt8, 700)
    ids = device_ids === nothing ? Cint[] : Cint.(device_ids)
    GC.@preserve ids err begin
        rc = ccall((:dfm_multi_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Cint, Ptr{Cint}, Cuint, Ptr{UInt8}, Cint),
                   ref, ngpu, device_ids === nothing ? Ptr{Cint}(C_NULL) : pointer(ids),
                   force_comm ? DFM_MULTI_F_FORCE_COMM : Cuint(0), err, length(err))
        rc == 0 || error("libdfmhip: dfm_multi_create: status $rc: $(unsafe_string(pointer(err)))")
    end
    m = Multi(ref[], (0, 0, 0, 0), 0)
    finalizer(x -> (x.ptr != C_NULL && ccall((:dfm_multi_destroy, LIB), Cint, (Ptr{Cvoid},), x.ptr); x.ptr = C_NULL), m)
    return m
end
const MULTIS = Dict{Int,Multi}()
function multi(ngpu::Integer = 1)
    m = get(MULTIS, Int(ngpu), nothing)
    (m === nothing || m.ptr == C_NULL) && (m = MULTIS[Int(ngpu)] = multi_create(ngpu))
    return m
end

"Upload a job: panel (N, T, B), Lam (r, N, B), R (N, B), A / Q / P0 (r, r, B), mu0 (r, B) -- the C layouts."
function multi_load(m::Multi, panel::Array{Float64,3}, Lam::Array{Float64,3}, R::Matrix{Float64}, A::Array{Float64,3},
                    Q::Array{Float64,3}, mu0::Matrix{Float64}, P0::Array{Float64,3})
    N, T, B = size(panel); r = size(Lam, 1)
    GC.@preserve panel Lam R A Q mu0 P0 begin
        mcheck(m, ccall((:dfm_multi_load, LIB), Cint,
                        (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), m.ptr, B, T, N, r, panel, Lam, R, A, Q, mu0, P0))
    end
    m.shape = (B, T, N, r)
    return m
end

"Generate a synthetic job where it lives (SURVEY 8(d) DGP): GPU g draws replicates first_replicate + [lo_g, hi_g); nothing
crosses PCIe (BASELINE configs[2]: 65 536 replicates).  pca_start: the PCA + OLS start instead of the DGP parameters."
function multi_synth(m::Multi, seed::Integer, first_replicate::Integer, B::Integer, T::Integer, N::Integer, r::Integer;
                     missing_prob::Real = 0.0, pca_start::Bool = false)
    mcheck(m, ccall((:dfm_multi_synth, LIB), Cint, (Ptr{Cvoid}, UInt64, Int64, Cint, Cint, Cint, Cint, Cdouble, Cint),
                    m.ptr, seed, first_replicate, B, T, N, r, missing_prob, pca_start ? 1 : 0))
    m.shape = (Int(B), Int(T), Int(N), Int(r))
    return m
end

function multi_ks_pass(m::Multi; want_P::Bool = true, flags::Cuint = Cuint(0))
    mcheck(m, ccall((:dfm_multi_ks_pass, LIB), Cint, (Ptr{Cvoid}, Cint, Cuint), m.ptr, want_P ? 1 : 0, flags))
end

"The EM loop on the resident job (one ncclAllGather of {loglik, active} per iteration); returns the iterations run."
function multi_em(m::Multi; max_iter::Integer = 50, tol::Real = 1e-6, want_smooth::Bool = false, want_P::Bool = false,
                  flags::Cuint = Cuint(0))
    ran = Ref{Cint}(0)
    m.max_iter = Int(max_iter)
    mcheck(m, ccall((:dfm_multi_em, LIB), Cint, (Ptr{Cvoid}, Cint, Cdouble, Cint, Cint, Cuint, Ptr{Cint}),
                    m.ptr, max_iter, tol, want_smooth ? 1 : 0, want_P ? 1 : 0, flags, ran))
    return Int(ran[])
end

"One resident array of the whole job in global replicate order, in its C layout (replicate index last in Julia)."
function multi_fetch(m::Multi, what::Cint)
    B, T, N, r = m.shape; np = div(r * (r + 1), 2)
    if what == MULTI_ITERS
        out = Array{Cint}(undef, B)
        GC.@preserve out mcheck(m, ccall((:dfm_multi_fetch, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}), m.ptr, what, pointer(out)))
        return out
    end
    dims = what == MULTI_LAM ? (r, N, B) : what == MULTI_R ? (N, B) : what == MULTI_MU0 ? (r, B) :
           what == MULTI_F_SMOOTH ? (r, T, B) : what == MULTI_P_SMOOTH ? (np, T, B) : what == MULTI_LOGLIK ? (B,) :
           what == MULTI_LOGLIK_PATH ? (m.max_iter, B) : what == MULTI_PANEL ? (N, T, B) : (r, r, B)
    out = Array{Float64}(undef, dims...)
    GC.@preserve out mcheck(m, ccall((:dfm_multi_fetch, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}), m.ptr, what, pointer(out)))
    return out
end

"EM for B replicates in ONE call on `ngpu` GPUs of this node: the cached multi-GPU object (one RCCL communicator per GPU
count for the whole session) -- replicates split over the GPUs, one host thread per GPU inside the library, ONE RCCL
all-gather of {loglik, active} after every EM iteration (SURVEY.md 8(e)).  panels[b]: T x N, starts[b] = (Lam, R, A, Q,
mu0, P0).  Returns the per-replicate estimates, log-likelihood paths and iteration counts."
function em_batch(panels::Vector{Matrix{Float64}}, starts::Vector; max_iter::Integer = 50, tol::Real = 1e-6,
                  ngpu::Integer = 1, want_smooth::Bool = false, singular_q::Bool = false)
    B = length(panels); T, N = size(panels[1]); r = size(starts[1].Lam, 2)
    panel = cat([permutedims(z, (2, 1)) for z in panels]...; dims = 3)
    Lam = pack3([p.Lam for p in starts]); R = hcat([p.R for p in starts]...)
    A = pack3([p.A for p in starts]); Q = pack3([p.Q for p in starts]); P0 = pack3([p.P0 for p in starts])
    mu0 = hcat([p.mu0 for p in starts]...)
    flags = (any(isnan, panel) ? DFM_F_MAY_HAVE_MISSING : Cuint(0)) | (singular_q ? DFM_F_SINGULAR_Q : Cuint(0))
    m = multi_load(multi(ngpu), panel, Lam, R, A, Q, mu0, P0)
    ran = multi_em(m; max_iter = max_iter, tol = tol, want_smooth = want_smooth, want_P = want_smooth, flags = flags)
    path = multi_fetch(m, MULTI_LOGLIK_PATH); iters = multi_fetch(m, MULTI_ITERS)
    Rm = multi_fetch(m, MULTI_R); mu = multi_fetch(m, MULTI_MU0)
    return (Lam = unpack3(multi_fetch(m, MULTI_LAM)), R = [Rm[:, b] for b in 1:B], A = unpack3(multi_fetch(m, MULTI_A)),
            Q = unpack3(multi_fetch(m, MULTI_Q)), mu0 = [mu[:, b] for b in 1:B], P0 = unpack3(multi_fetch(m, MULTI_P0)),
            loglik = [path[1:iters[b], b] for b in 1:B], iters = Int.(iters), iterations = ran,
            factor = want_smooth ? unpack3(multi_fetch(m, MULTI_F_SMOOTH)) : nothing)
end

"EM for VAR(p) factor dynamics in companion form (dfm_em_varp_batch; include/dfm_hip.h): p.Avar is r x (r p) =
[A_1 .. A_p], p.mu0 / p.P0 the moments of z_0 = (f_0, .., f_{1-p})."
function em_varp(h::Handle, z::Matrix{Float64}, p, nlag::Integer; max_iter::Integer = 50, tol::Real = 1e-6,
                 singular_q::Bool = false)   # singular_q: the r x r block Q may be rank deficient (DFM_F_SINGULAR_Q)
    T, N = size(z); r = size(p.Lam, 2); k = r * nlag
    panel = to_c_panel(z)
    Lam = reshape(permutedims(p.Lam), r, N, 1); R = reshape(copy(p.R), N, 1)
    Avar = reshape(permutedims(p.Avar), k, r, 1); Q = reshape(permutedims(p.Q), r, r, 1)
    mu0 = reshape(copy(p.mu0), k, 1); P0 = reshape(permutedims(p.P0), k, k, 1)
    path = Array{Float64}(undef, max_iter, 1); iters = Array{Cint}(undef, 1)
    f = Array{Float64}(undef, r, T, 1); np = div(r * (r + 1), 2); P = Array{Float64}(undef, np, T, 1)
    flags = (any(isnan, z) ? DFM_F_MAY_HAVE_MISSING : Cuint(0)) | (singular_q ? DFM_F_SINGULAR_Q : Cuint(0))
    GC.@preserve panel Lam R Avar Q mu0 P0 path iters f P begin
        rc = ccall((:dfm_em_varp_batch, LIB), Cint,
                   (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Cdouble, Ptr{Float64}, Ptr{Cint},
                    Ptr{Float64}, Ptr{Float64}, Cuint),
                   h.ptr, 1, T, N, r, nlag, panel, Lam, R, Avar, Q, mu0, P0, max_iter, tol, path, iters, f, P, flags)
        check(h.ptr, rc)
    end
    kk = Int(iters[1])
    return (Lam = permutedims(Lam[:, :, 1]), R = R[:, 1], A = permutedims(Avar[:, :, 1]), Q = permutedims(Q[:, :, 1]),
            mu0 = mu0[:, 1], P0 = permutedims(P0[:, :, 1]), loglik = path[1:kk, 1], iters = kk,
            factor = permutedims(f[:, :, 1]))
end

"Smoother pass with AR(q) idiosyncratic terms (dfm_ks_pass_ar_batch; include/dfm_hip.h): x is T x N (NaN = missing, in
deviations from its intercept), Lam N x r, sig2 = uar_ser.^2, rho = uar_coef (N x q), Avar r x (r p), Q r x r, mu0 / P0
the moments of z_q, r max(p, q+1) wide.  Returns the smoothed factors of rows q+1..T and the conditional log-likelihood."
function ks_pass_ar(h::Handle, x::Matrix{Float64}, Lam::Matrix{Float64}, sig2::Vector{Float64}, rho::Matrix{Float64},
                    Avar::Matrix{Float64}, Q::Matrix{Float64}, mu0::Vector{Float64}, P0::Matrix{Float64};
                    singular_q::Bool = false)
    T, N = size(x); r = size(Lam, 2); q = size(rho, 2); p = div(size(Avar, 2), r)
    panel = to_c_panel(x)
    LamC = permutedims(Lam); rhoC = permutedims(rho); AC = permutedims(Avar); QC = permutedims(Q); P0C = permutedims(P0)
    f = Array{Float64}(undef, r, T - q); np = div(r * (r + 1), 2); P = Array{Float64}(undef, np, T - q)
    ll = Array{Float64}(undef, 1)
    flags = (any(isnan, x) ? DFM_F_MAY_HAVE_MISSING : Cuint(0)) | (singular_q ? DFM_F_SINGULAR_Q : Cuint(0))
    GC.@preserve panel LamC sig2 rhoC AC QC mu0 P0C f P ll begin
        rc = ccall((:dfm_ks_pass_ar_batch, LIB), Cint,
                   (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cuint),
                   h.ptr, 1, T, N, r, p, q, panel, LamC, sig2, rhoC, AC, QC, mu0, P0C, f, P, ll, flags)
        check(h.ptr, rc)
    end
    return (factor = permutedims(f), P = permutedims(P), loglik = ll[1])
end

"Joint ECM estimation with AR(q) idiosyncratic terms (dfm_em_ar_batch; include/dfm_hip.h): the re-estimation of the
reference's `lambda`, `uar_coef`, `uar_ser` (dfm_functions.ipynb:391-415) and factor VAR inside the parametric model.
Arguments as ks_pass_ar (the start: what `estimate!(m)` left in the model).  Returns the updated parameters, the
log-likelihood path (conditional on the first q rows; non-decreasing) and the smoothed factors of rows q+1..T."
function em_ar(h::Handle, x::Matrix{Float64}, Lam::Matrix{Float64}, sig2::Vector{Float64}, rho::Matrix{Float64},
               Avar::Matrix{Float64}, Q::Matrix{Float64}, mu0::Vector{Float64}, P0::Matrix{Float64};
               max_iter::Integer = 20, tol::Real = 1e-6, singular_q::Bool = false)
    T, N = size(x); r = size(Lam, 2); q = size(rho, 2); p = div(size(Avar, 2), r)
    panel = to_c_panel(x)
    LamC = permutedims(Lam); sig = copy(sig2); rhoC = permutedims(rho); AC = permutedims(Avar); QC = permutedims(Q)
    mu = copy(mu0); P0C = permutedims(P0)
    path = Array{Float64}(undef, max_iter); iters = Array{Cint}(undef, 1)
    f = Array{Float64}(undef, r, T - q); np = div(r * (r + 1), 2); P = Array{Float64}(undef, np, T - q)
    flags = (any(isnan, x) ? DFM_F_MAY_HAVE_MISSING : Cuint(0)) | (singular_q ? DFM_F_SINGULAR_Q : Cuint(0))
    GC.@preserve panel LamC sig rhoC AC QC mu P0C path iters f P begin
        rc = ccall((:dfm_em_ar_batch, LIB), Cint,
                   (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Cdouble, Ptr{Float64}, Ptr{Cint},
                    Ptr{Float64}, Ptr{Float64}, Cuint),
                   h.ptr, 1, T, N, r, p, q, panel, LamC, sig, rhoC, AC, QC, mu, P0C, max_iter, tol, path, iters, f, P, flags)
        check(h.ptr, rc)
    end
    kk = Int(iters[1])
    return (Lam = permutedims(LamC), sig2 = sig, rho = permutedims(rhoC), Avar = permutedims(AC), Q = permutedims(QC),
            mu0 = mu, P0 = permutedims(P0C), loglik = path[1:kk], iters = kk, factor = permutedims(f))
end

# ---- the reference's NON-parametric estimator on the GPU (als.hip) ------------------------------------------------
"`estimate_factor!` sweeps (dfm_functions.ipynb:352-370) for ONE run: z is the standardised T x N window (NaN =
missing), F0 the T x r start (pca_score).  dfm_als_batch; returns factors, loadings (NaN rows: no loadings), ssr,
iterations, R2."
function als(h::Handle, z::Matrix{Float64}, F0::Matrix{Float64}; nt_min::Integer = 20,
             max_iter::Integer = 100000000, tol::Real = 1e-8, want_R2::Bool = true)
    T, N = size(z); r = size(F0, 2)
    zc = permutedims(z, (2, 1))                                   # (N, T) column-major == C [t][i]
    F = reshape(permutedims(F0, (2, 1)), r, T, 1)                 # C [b][t][k]
    Lam = Array{Float64}(undef, r, N, 1); iters = Array{Cint}(undef, 1); ssr = Array{Float64}(undef, 1)
    R2 = Array{Float64}(undef, N, 1)
    GC.@preserve zc F Lam iters ssr R2 begin
        rc = ccall((:dfm_als_batch, LIB), Cint,
                   (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ptr{Float64}, Clonglong, Ptr{Cint}, Ptr{Float64}, Ptr{Float64},
                    Cint, Cint, Cdouble, Ptr{Float64}, Cint, Ptr{Cint}, Ptr{Float64}, Ptr{Float64}),
                   h.ptr, 1, T, N, r, zc, 0, C_NULL, F, Lam, nt_min, min(max_iter, typemax(Cint)), tol, C_NULL, 0,
                   iters, ssr, want_R2 ? pointer(R2) : Ptr{Float64}(C_NULL))
        check(h.ptr, rc)
    end
    return (factor = permutedims(F[:, :, 1]), Lam = permutedims(Lam[:, :, 1]), ssr = ssr[1], iters = Int(iters[1]),
            R2 = R2[:, 1])
end

"`estimate_factor!` sweeps for B runs in ONE call (dfm_als_batch): the runs of `estimate_factor_numbers` /
`amengual_watson_test` (dfm_functions.ipynb:698-768).  zs: one T x N window shared by every run, or a vector of B windows;
F0s[b]: T x r_b start (the first r_b columns of a `pca_score`); every run may have its own number of factors."
function als_batch(h::Handle, zs, F0s::Vector{Matrix{Float64}}; nt_min::Integer = 20, max_iter::Integer = 100000000,
                   tol::Real = 1e-8)
    shared = zs isa AbstractMatrix
    B = length(F0s); T = size(F0s[1], 1); r = maximum(size(f, 2) for f in F0s)
    N = shared ? size(zs, 2) : size(zs[1], 2)
    zc = shared ? permutedims(zs, (2, 1)) : cat([permutedims(z, (2, 1)) for z in zs]...; dims = 3)   # C [b][t][i]
    F = zeros(r, T, B)
    for b in 1:B
        F[1:size(F0s[b], 2), :, b] = permutedims(F0s[b], (2, 1))
    end
    r_each = Cint[size(f, 2) for f in F0s]
    Lam = Array{Float64}(undef, r, N, B); iters = Array{Cint}(undef, B); ssr = Array{Float64}(undef, B)
    R2 = Array{Float64}(undef, N, B)
    GC.@preserve zc r_each F Lam iters ssr R2 begin
        rc = ccall((:dfm_als_batch, LIB), Cint,
                   (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ptr{Float64}, Clonglong, Ptr{Cint}, Ptr{Float64}, Ptr{Float64},
                    Cint, Cint, Cdouble, Ptr{Float64}, Cint, Ptr{Cint}, Ptr{Float64}, Ptr{Float64}),
                   h.ptr, B, T, N, r, zc, shared ? 0 : T * N, r_each, F, Lam, nt_min, min(max_iter, typemax(Cint)), tol,
                   C_NULL, 0, iters, ssr, R2)
        check(h.ptr, rc)
    end
    return (factor = [permutedims(F[1:r_each[b], :, b]) for b in 1:B], Lam = [permutedims(Lam[1:r_each[b], :, b]) for b in 1:B],
            ssr = ssr, iters = Int.(iters), R2 = [R2[:, b] for b in 1:B])
end

"P complete-case regressions (`ols_skipmissing(..., Balanced())`, dfm_functions.ipynb:242-252): column p of Y (T x P,
NaN = missing) on the shared regressors X (T x K).  dfm_ols_batch; returns beta (K x P), resid (T x P), ssr, tss, nobs."
function ols(h::Handle, X::Matrix{Float64}, Y::Matrix{Float64}; nt_min::Integer = 0)
    T, K = size(X); P = size(Y, 2)
    Xc = permutedims(X, (2, 1)); Yc = permutedims(Y, (2, 1))      # C [t][k], C [t][p]
    beta = Array{Float64}(undef, K, P); resid = Array{Float64}(undef, T, P)
    ssr = Array{Float64}(undef, P); tss = similar(ssr); nobs = Array{Cint}(undef, P)
    GC.@preserve Xc Yc beta resid ssr tss nobs begin
        rc = ccall((:dfm_ols_batch, LIB), Cint,
                   (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Float64}, Clonglong, Ptr{Float64}, Clonglong, Clonglong, Cint,
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cint}),
                   h.ptr, P, T, K, Xc, 0, Yc, 1, P, nt_min, beta, resid, ssr, tss, nobs)
        check(h.ptr, rc)
    end
    return (beta = beta, resid = resid, ssr = ssr, tss = tss, nobs = Int.(nobs))   # beta[:, p]: C [p][k] == Julia (k, p)
end

"Chow statistics with HAC covariance (dfm_chow_batch): series s = complete cases ys[s] (vector), Xs[s] (T_s x k);
problem p = (prob_series[p] (1-based), prob_break[p] rows before the break, bandwidth prob_q[p]).  `compute_qlr` is the
maximum over the problems of one series (dfm_functions.ipynb:1019-1047)."
function chow_batch(h::Handle, ys::Vector{Vector{Float64}}, Xs::Vector{Matrix{Float64}}, prob_series::Vector{<:Integer},
                    prob_break::Vector{<:Integer}, prob_q::Vector{<:Integer})
    S = length(ys); k = size(Xs[1], 2); Tlen = Cint[length(v) for v in ys]; Tmax = maximum(Tlen)
    y = zeros(Tmax, S); X = zeros(k, Tmax, S)                     # C [s][t], C [s][t][k]
    for s in 1:S
        y[1:Tlen[s], s] = ys[s]; X[:, 1:Tlen[s], s] = permutedims(Xs[s], (2, 1))
    end
    ps = Cint.(prob_series .- 1); pb = Cint.(prob_break); pq = Cint.(prob_q); P = length(ps)
    out = Array{Float64}(undef, P)
    GC.@preserve y X Tlen ps pb pq out begin
        rc = ccall((:dfm_chow_batch, LIB), Cint,
                   (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Cint}, Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Cint},
                    Ptr{Float64}), h.ptr, S, Tmax, k, y, X, Tlen, P, ps, pb, pq, out)
        check(h.ptr, rc)
    end
    return out
end

"B wild-bootstrap draws of the VAR's impulse responses (dfm_var_bootstrap_irf) and their nearest-rank quantile bands
(dfm_quantile_bands).  y, resid: T x ns over the estimation window; betahat: (1 + ns p) x ns as in `estimate_var!`."
function bootstrap_irf(h::Handle, y::Matrix{Float64}, betahat::Matrix{Float64}, resid::Matrix{Float64}, p::Integer,
                       H::Integer, ndraws::Integer; seed::Integer = 20160415, first_draw::Integer = 0,
                       quantiles = [0.05, 0.16, 0.5, 0.84, 0.95])
    T, ns = size(y)
    yc = permutedims(y, (2, 1)); bc = permutedims(betahat, (2, 1)); ec = permutedims(resid, (2, 1))
    irf = Array{Float64}(undef, ns, H, ns, ndraws)                # C [d][i][h][k] == Julia (k, h, i, d)
    q = Float64.(quantiles); bands = Array{Float64}(undef, ns, H, ns, length(q))
    GC.@preserve yc bc ec irf q bands begin
        rc = ccall((:dfm_var_bootstrap_irf, LIB), Cint,
                   (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                    UInt64, Int64, Ptr{Float64}, Ptr{Float64}),
                   h.ptr, ndraws, T, ns, p, H, yc, bc, ec, C_NULL, seed, first_draw, C_NULL, irf)
        check(h.ptr, rc)
        rc = ccall((:dfm_quantile_bands, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                   h.ptr, ndraws, ns * H * ns, length(q), irf, q, bands)
        check(h.ptr, rc)
    end
    # to the reference's irf[variable, horizon, shock] (dfm_functions.ipynb:793-816), draws / quantiles last
    return (draws = permutedims(irf, (3, 2, 1, 4)), bands = permutedims(bands, (3, 2, 1, 4)))
end

end # module

import Random
using LinearAlgebra

# ---------------------------------------------------------------------------------------------------------
# The new method (SURVEY.md 8(b)).  Same mutate-in-place convention as the reference's estimate!
# (dfm_functions.ipynb:530-543); additionally returns the per-iteration log-likelihood vector.
#   nrep > 0: after the point estimate, `nrep` parametric-bootstrap replicates of the standardised window are drawn
#   from the fitted model (same missing pattern; Julia's RNG seeded with `seed`) and re-estimated by EM IN ONE BATCHED
#   CALL on `ngpu` GPUs (DFMHip.em_batch -> dfm_em_batch_multi: replicates sharded over the GPUs, one RCCL all-gather
#   per EM iteration); the method then returns (loglik = ..., replicates = ...).
function estimate!(m::DFMModel, ::Parametric; max_em_iter::Integer = 50, tol_em::Real = 1e-6,
                   factor_lags::Integer = m.n_factorlag, nrep::Integer = 0, seed::Integer = 20160415, ngpu::Integer = 1,
                   device::Integer = 0, handle = nothing, lam_constr_f = nothing, lam_constr_fl = nothing)
    (lam_constr_f === nothing && lam_constr_fl === nothing) ||
        error("loading constraints are not supported on the parametric path")   # never drop them silently
    m.nfac_o == 0 || return estimate_observed!(m; max_em_iter = max_em_iter, tol_em = tol_em, device = device, handle = handle)
    r = m.nfac_u
    nlag = Int(factor_lags)
    (nlag >= 1 && r * nlag <= 32) || error("need 1 <= factor_lags and nfac_u * factor_lags <= 32")
    (nrep == 0 || nlag == 1) || error("bootstrap replicates (nrep > 0) need factor_lags = 1")
    incl = m.inclcode .== 1
    xdata = m.data[m.initperiod:m.lastperiod, incl]                       # dfm_functions.ipynb:335-336
    xstd, xsd = standardize_data(xdata)                                   # :339
    m.fes.tss = sum(skipmissing(xstd .^ 2))                               # :342
    m.fes.nobs = count(.!ismissing.(xstd))                                # :343
    # :357 -- a series with fewer than nt_min_factor_estimation observed periods gets no loadings in the reference's
    # estimator: it is left out here too (api.estimate: `enough`)
    enough = vec(sum(.!ismissing.(xstd), dims = 1)) .>= m.nt_min_factor_estimation
    xstd = xstd[:, enough]; xsd = vec(xsd)[enough]
    z = reshape(DFMHip.nan_for_missing(xstd), size(xstd))
    xbal, balmask = drop_missing_col(xstd)                                # :345
    balmask = vec(balmask)
    size(xbal, 2) >= r || error("fewer fully observed series than factors: cannot initialise by PCA")
    h = handle === nothing ? DFMHip.handle(device) : handle               # one cached handle per device
    p0 = DFMHip.pca_init(h, Float64.(xbal), r)                            # pca_score (:179-183) + OLS start, on the GPU
    N = size(z, 2)
    Lam = Matrix{Float64}(undef, N, r); R = Vector{Float64}(undef, N)
    Lam[balmask, :] = p0.Lam; R[balmask] = p0.R
    gap = findall(.!balmask)
    if !isempty(gap)                                                      # series with gaps: complete-case OLS on the PCA
        o = DFMHip.ols(h, p0.F, z[:, gap])                                # factors (:242-252), one dfm_ols_batch call
        Lam[gap, :] = permutedims(o.beta); R[gap] = o.ssr ./ max.(o.nobs, 1)
    end
    fit = if nlag == 1
        DFMHip.em(h, z, (Lam = Lam, R = R, A = p0.A, Q = p0.Q, mu0 = p0.mu0, P0 = p0.P0);
                  max_iter = max_em_iter, tol = tol_em)
    else   # VAR(p) start: OLS of the PCA factors on their lags, no constant (dfm_ols_batch); companion-form EM
        T = size(z, 1)
        Z = hcat([p0.F[nlag-l:T-l, :] for l in 0:nlag-1]...)                  # row j: (f_{p-1+j}, .., f_j)
        o = DFMHip.ols(h, Z[1:end-1, :], p0.F[nlag+1:end, :])
        Qv = o.resid' * o.resid / (T - nlag)
        P0v = Z' * Z / size(Z, 1)
        DFMHip.em_varp(h, z, (Lam = Lam, R = R, Avar = permutedims(o.beta), Q = (Qv + Qv') / 2, mu0 = zeros(r * nlag),
                              P0 = (P0v + P0v') / 2), nlag; max_iter = max_em_iter, tol = tol_em)
    end
    m.factor[m.initperiod:m.lastperiod, :] = fit.factor                   # in place: aliases factor_var_model.y (:80, :371)
    cols = findall(incl)[enough]
    m.lambda[cols, :] = fit.Lam .* xsd
    m.uar_ser[cols] = sqrt.(fit.R) .* xsd
    m.uar_coef[cols, :] .= 0.0
    common = fit.factor * fit.Lam'
    e = [isnan(z[t, i]) ? 0.0 : z[t, i] - common[t, i] for t in 1:size(z, 1), i in 1:N]
    m.fes.ssr = sum(abs2, e)                                              # :366
    var = m.factor_var_model                                              # fill_matrices! (:477-492), VAR(1) block
    fill!(var.M, 0.0); fill!(var.Q, 0.0); fill!(var.G, 0.0)
    ka = min(size(fit.A, 2), size(var.M, 2))
    var.M[1:r, 1:ka] = fit.A[:, 1:ka]                                     # [A_1 .. A_p] (:484-486)
    var.nlag > 1 && (var.M[r+1:end, 1:end-r] = Matrix(1.0I, r * (var.nlag - 1), r * (var.nlag - 1)))
    var.Q[1:r, 1:r] = Matrix(1.0I, r, r)
    var.seps[:, :] = fit.Q
    # fill_matrices! takes the lower Cholesky factor (:489); a rank-deficient Q (covariance-form fit) has none: its
    # symmetric square root then stands in (G G' = Q still holds)
    var.G[1:r, 1:r] = isposdef(Symmetric(fit.Q)) ? cholesky(Symmetric(fit.Q)).L :
                      (e = eigen(Symmetric(fit.Q)); e.vectors * Diagonal(sqrt.(max.(e.values, 0.0))) * e.vectors')
    nrep == 0 && return fit.loglik
    # ---- parametric-bootstrap replicates, re-estimated in one batched multi-GPU call (api.estimate: same steps) ----
    rng = Random.MersenneTwister(seed)
    T = size(z, 1)
    # square roots by eigendecomposition: a fit that needed the covariance-form recursion (DFM_F_SINGULAR_Q) has a
    # rank-deficient Q, and `cholesky` would throw AFTER the point estimate was written into the model
    psd_sqrt(S) = (e = eigen(Symmetric((S + S') / 2)); e.vectors * Diagonal(sqrt.(max.(e.values, 0.0))))
    LQ = psd_sqrt(fit.Q)
    LS = psd_sqrt(fit.P0)                                                 # start the factor path from the fitted f_0 moments
    panels = Vector{Matrix{Float64}}(undef, nrep)
    for b in 1:nrep
        f = fit.mu0 + LS * randn(rng, r)
        x = Matrix{Float64}(undef, T, N)
        for t in 1:T
            f = fit.A * f + LQ * randn(rng, r)
            x[t, :] = fit.Lam * f + sqrt.(fit.R) .* randn(rng, N)
        end
        x[isnan.(z)] .= NaN                                               # the window's own missing pattern
        panels[b] = x
    end
    start = (Lam = fit.Lam, R = fit.R, A = fit.A, Q = fit.Q, mu0 = fit.mu0, P0 = fit.P0)
    reps = DFMHip.em_batch(panels, [start for _ in 1:nrep]; max_iter = max_em_iter, tol = tol_em, ngpu = ngpu,
                           singular_q = fit.singular_q)
    return (loglik = fit.loglik, replicates = reps)
end


# ---------------------------------------------------------------------------------------------------------
# The reference's own estimator on the GPU: same arguments and effects as `estimate_factor!(m, max_iter, computeR2)`
# (dfm_functions.ipynb:328-382), PCA start and every sweep on the GPU.  Loading constraints (`lam_constr`, used at
# Stock_Watson.ipynb:1336-1344) are NOT implemented on the HIP path: with constraints this method hands the call to the
# reference's own pure-Julia method when the maintainer has kept it under the name `estimate_factor_cpu!` (see
# INTEGRATION.md for the two-line change), and raises otherwise -- it never drops them.
function estimate_factor_hip!(m::DFMModel, max_iter::Integer = 100000000, computeR2::Bool = true; lam_constr = nothing,
                              device::Integer = 0, handle = nothing)
    if lam_constr !== nothing
        isdefined(Main, :estimate_factor_cpu!) ||
            error("estimate_factor_hip!: loading constraints are not supported on the HIP path and the reference's " *
                  "method is not available as estimate_factor_cpu! (INTEGRATION.md)")
        return Main.estimate_factor_cpu!(m, max_iter, computeR2; lam_constr = lam_constr)
    end
    m.nfac_o == 0 || error("observed factors are not supported on the HIP path")
    xdata = m.data[m.initperiod:m.lastperiod, m.inclcode .== 1]           # :335-336
    xstd, _ = standardize_data(xdata)                                     # :339
    m.fes.tss = sum(skipmissing(xstd .^ 2))                               # :342
    m.fes.nobs = count(.!ismissing.(xstd))                                # :343
    xbal, _ = drop_missing_col(xstd)                                      # :345
    h = handle === nothing ? DFMHip.handle(device) : handle               # cached: the notebook calls this ~200 times a table
    F0 = DFMHip.pca_init(h, Float64.(xbal), m.nfac_u).F                  # pca_score (:348)
    z = reshape(DFMHip.nan_for_missing(xstd), size(xstd))
    fit = DFMHip.als(h, z, F0; nt_min = m.nt_min_factor_estimation, max_iter = max_iter, tol = m.tol,
                     want_R2 = computeR2)                                 # :352-370, :372-380
    m.factor[m.initperiod:m.lastperiod, :] = fit.factor                   # :371
    m.fes.ssr = fit.ssr                                                   # :366
    computeR2 && (m.fes.R2 = fit.R2)
    return nothing
end

# `impulse_response(varm, shock_id::Real, T)` (dfm_functions.ipynb:817-821) repaired.  The reference's method cannot run: it passes an
# undefined `x` and SIX arguments to the five-argument `compute_irf_single_shock!(irfs, varm, i, shock_id, T)` (:801-810).  What it
# means is evident from the vector method (:793-799): the same recursion -- irf[:, t] = Q M^(t-1) G[:, shock_id] -- written into a
# ny x T matrix (`irfs[:, t, 1]` addresses a Matrix: a trailing index 1 is legal).  Defining it here REPLACES the broken method once
# the notebook's functions are loaded (same signature); the vector and `:all` methods (:793-799, :822-825) are the reference's own.
# Mirror: dynamic_factor_models_amd/api.py impulse_response (scalar shock id).
function impulse_response(varm::VARModel, shock_id::Real, T::Integer)
    isinteger(shock_id) && 1 <= shock_id <= size(varm.G, 2) ||
        throw(ArgumentError("shock_id must be an integer in 1:$(size(varm.G, 2))"))
    irfs = Matrix{Float64}(undef, size(varm.Q, 1), T)
    compute_irf_single_shock!(irfs, varm, 1, Int(shock_id), T)
    return irfs
end

# Wild-bootstrap bands of `impulse_response(varm, shock_ids, T)` (dfm_functions.ipynb:793-816) for an estimated VARModel.
function bootstrap_irf_bands(varm::VARModel, H::Integer; ndraws::Integer = 10000, seed::Integer = 20160415,
                             device::Integer = 0, handle = nothing)
    rows = findall(t -> !any(ismissing, varm.resid[t, :]), 1:size(varm.resid, 1))
    first = rows[1] - varm.nlag
    y = Float64.(varm.y[first:rows[end], :])
    resid = zeros(size(y)); resid[varm.nlag+1:end, :] = Float64.(varm.resid[rows, :])
    h = handle === nothing ? DFMHip.handle(device) : handle
    return DFMHip.bootstrap_irf(h, y, Float64.(varm.betahat), resid, varm.nlag, H, ndraws; seed = seed)
end

# ---------------------------------------------------------------------------------------------------------
# `estimate_factor_numbers(m, nfacs)` (dfm_functions.ipynb:698-725) with its batch axis on the GPU.  The reference runs
# `estimate_factor!` once per static factor count and, inside each, `amengual_watson_test` (:734-768) runs it once more per
# dynamic factor count: 10 + 55 serial estimations for Table 2 (Stock_Watson.ipynb:516, 591, 643, 893).  Here:
#   1 dfm_pca_init_batch + ONE dfm_als_batch        for the max_nfac static runs (shared window, r_each = 1..max_nfac;
#                                                    `pca_score` columns are nested, so run r starts from the first r);
#   per static count i: 1 dfm_ols_batch             (every series on [1, lags of the i factors], :741-750) and
#                       1 dfm_pca_init_batch        (start of the residual panel's runs);
#   ONE dfm_als_batch                                for ALL max_nfac (max_nfac + 1) / 2 dynamic runs, each on its own
#                                                    residual window.
# Same return type and fields as the reference (FactorNumberEstimateStats); mirrors api.estimate_factor_numbers(with_aw).
function estimate_factor_numbers_hip(m::DFMModel, nfacs::Union{Real, AbstractVector}; device::Integer = 0, handle = nothing)
    m.nfac_o == 0 || error("observed factors are not supported on the HIP path")
    max_nfac = Int(maximum(nfacs))
    h = handle === nothing ? DFMHip.handle(device) : handle
    incl = m.inclcode .== 1
    ns = count(incl)
    xdata = m.data[m.initperiod:m.lastperiod, incl]                       # :335-336
    xstd, _ = standardize_data(xdata)                                     # :339
    tss = sum(skipmissing(xstd .^ 2)); nobs = count(.!ismissing.(xstd)); T = size(xstd, 1)
    xbal, _ = drop_missing_col(xstd)                                      # :345
    F0 = DFMHip.pca_init(h, Float64.(xbal), max_nfac).F                   # pca_score (:348), nested columns
    z = reshape(DFMHip.nan_for_missing(xstd), size(xstd))
    st = DFMHip.als_batch(h, z, [F0[:, 1:k] for k in 1:max_nfac]; nt_min = m.nt_min_factor_estimation, tol = m.tol)
    bn(ssr, no, Tn, k) = (nbar = no / Tn; log(ssr / no) + k * log(min(nbar, Tn)) * (nbar + Tn) / no)   # bai_ng_criterion (:684-690)
    bn_icp = Vector{Union{Missing, Float64}}(undef, max_nfac)
    ssr_static = Vector{Float64}(undef, max_nfac)
    R2_static = Matrix{Union{Missing, Float64}}(undef, ns, max_nfac)
    aw_icp = Matrix{Union{Missing, Float64}}(missing, max_nfac, max_nfac)
    ssr_dynamic = Matrix{Union{Missing, Float64}}(missing, max_nfac, max_nfac)
    R2_dynamic = Array{Union{Missing, Float64}}(missing, ns, max_nfac, max_nfac)
    nan2missing(v) = Union{Missing, Float64}[isnan(x) ? missing : x for x in v]
    for i in 1:max_nfac
        bn_icp[i] = bn(st.ssr[i], nobs, T, i); ssr_static[i] = st.ssr[i]; R2_static[:, i] = nan2missing(st.R2[i])
    end
    # ---- amengual_watson_test of every static run: residual panels, then ALL dynamic runs in one call ----
    est = m.data[:, incl]
    Tall = size(est, 1)
    nlag = m.factor_var_model.nlag
    estn = reshape(DFMHip.nan_for_missing(est), size(est))
    zs = Matrix{Float64}[]; F0s = Matrix{Float64}[]; owner = Tuple{Int,Int}[]; nobs_i = Int[]; T_i = Int[]
    for i in 1:max_nfac
        fac = Matrix{Union{Missing, Float64}}(missing, Tall, i)
        fac[m.initperiod:m.lastperiod, :] = st.factor[i]
        x = [ones(Tall) lagmat(fac, 1:nlag)]                              # :741
        xn = reshape(DFMHip.nan_for_missing(x), size(x))
        # a series keeps its residuals when it has at least nt_min rows MORE than regressors (:744)
        o = DFMHip.ols(h, xn, estn; nt_min = size(xn, 2) + m.nt_min_factor_estimation)
        res = Union{Missing, Float64}[isnan(v) ? missing : v for v in o.resid]
        rstd, _ = standardize_data(res[m.initperiod+4:m.lastperiod, :])  # :761 (the reference hard-codes the 4)
        rbal, _ = drop_missing_col(rstd)
        Fi = DFMHip.pca_init(h, Float64.(rbal), i).F
        zi = reshape(DFMHip.nan_for_missing(rstd), size(rstd))
        for k in 1:i
            push!(zs, zi); push!(F0s, Fi[:, 1:k]); push!(owner, (k, i))
        end
        push!(nobs_i, count(.!ismissing.(rstd))); push!(T_i, size(rstd, 1))
    end
    dy = DFMHip.als_batch(h, zs, F0s; nt_min = m.nt_min_factor_estimation, tol = m.tol)
    for (b, (k, i)) in enumerate(owner)
        aw_icp[k, i] = bn(dy.ssr[b], nobs_i[i], T_i[i], k)
        ssr_dynamic[k, i] = dy.ssr[b]
        R2_dynamic[:, k, i] = nan2missing(dy.R2[b])
    end
    return FactorNumberEstimateStats(bn_icp, ssr_static, R2_static, aw_icp, ssr_dynamic, R2_dynamic, tss, nobs, T)
end


# ---------------------------------------------------------------------------------------------------------
# `estimate!(m, Parametric())` with OBSERVED factors (nfac_o > 0).  The reference's estimator is non-functional there
# (dfm_functions.ipynb:358-359, :371), so the semantics are those its data layout implies (include/dfm_hip.h,
# oracle/obs_oracle.py; api._estimate_parametric_observed is the same sequence): the caller has filled
# m.factor[initperiod:lastperiod, 1:nfac_o] with the observed factors; they are known regressors of the measurement
# equation, only the nfac_u remaining factors are latent.  Afterwards m.factor[:, nfac_o+1:end] = E[f_t | X], m.lambda the
# nfac_t loadings, and the factor VAR of ALL factors is the reference's own `estimate_var!(m.factor_var_model)`.
function estimate_observed!(m::DFMModel; max_em_iter::Integer = 50, tol_em::Real = 1e-6, device::Integer = 0, handle = nothing)
    ro, ru = m.nfac_o, m.nfac_u
    incl = m.inclcode .== 1
    Gm = m.factor[m.initperiod:m.lastperiod, 1:ro]
    any(ismissing, Gm) && error("observed factors: fill m.factor[initperiod:lastperiod, 1:nfac_o] (no gaps) before estimate!")
    G = Float64.(Gm)
    xstd, xsd = standardize_data(m.data[m.initperiod:m.lastperiod, incl])      # :335-339
    m.fes.tss = sum(skipmissing(xstd .^ 2)); m.fes.nobs = count(.!ismissing.(xstd))
    enough = vec(sum(.!ismissing.(xstd), dims = 1)) .>= m.nt_min_factor_estimation
    xstd = xstd[:, enough]; xsd = vec(xsd)[enough]
    z = reshape(DFMHip.nan_for_missing(xstd), size(xstd))
    h = handle === nothing ? DFMHip.handle(device) : handle
    og = DFMHip.ols(h, G, z)                                                   # every series on the observed factors
    res = Union{Missing, Float64}[isnan(v) ? missing : v for v in og.resid]
    rbal, balmask = drop_missing_col(res)
    balmask = vec(balmask)
    size(rbal, 2) >= ru || error("fewer fully observed series than unobserved factors: cannot initialise by PCA")
    p0 = DFMHip.pca_init(h, Float64.(rbal), ru)
    N = size(z, 2)
    Lam_u = Matrix{Float64}(undef, N, ru); R = Vector{Float64}(undef, N)
    Lam_u[balmask, :] = p0.Lam; R[balmask] = p0.R
    gap = findall(.!balmask)
    if !isempty(gap)
        o = DFMHip.ols(h, p0.F, og.resid[:, gap])
        Lam_u[gap, :] = permutedims(o.beta); R[gap] = o.ssr ./ max.(o.nobs, 1)
    end
    fit = DFMHip.em_obs(h, z, G, (Lam = hcat(permutedims(og.beta), Lam_u), R = R, A = p0.A, Q = p0.Q, mu0 = p0.mu0, P0 = p0.P0);
                        max_iter = max_em_iter, tol = tol_em)
    m.factor[m.initperiod:m.lastperiod, ro+1:end] = fit.factor               # in place (aliases factor_var_model.y, :80)
    cols = findall(incl)[enough]
    m.lambda[cols, :] = fit.Lam .* xsd
    m.uar_ser[cols] = sqrt.(fit.R) .* xsd
    m.uar_coef[cols, :] .= 0.0
    common = hcat(G, fit.factor) * fit.Lam'
    m.fes.ssr = sum(abs2, [isnan(z[t, i]) ? 0.0 : z[t, i] - common[t, i] for t in 1:size(z, 1), i in 1:N])
    estimate_var!(m.factor_var_model)                                          # VAR of (g, f): the reference's second stage (:444-492)
    return fit.loglik
end
