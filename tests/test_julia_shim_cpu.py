"""CPU check of the Julia side of the boundary: every `ccall` in julia/dfm_hip.jl is parsed and compared with the
prototype of the same symbol in include/dfm_hip.h -- symbol name, return type, argument count and the C type behind
every Julia argument type.  (Julia is not installed in the build image, so the shim cannot be executed there; the
Python binding table is checked the same way in test_cabi_and_api_cpu.py.)"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# C type (normalised) -> the Julia ccall argument types that are ABI-compatible with it
C2J = {
    "dfm_handle*": {"Ptr{Cvoid}"},
    "dfm_handle**": {"Ref{Ptr{Cvoid}}", "Ptr{Ptr{Cvoid}}"},
    "dfm_multi*": {"Ptr{Cvoid}"},
    "dfm_multi**": {"Ref{Ptr{Cvoid}}", "Ptr{Ptr{Cvoid}}"},
    "void*": {"Ptr{Cvoid}"},
    "int": {"Cint"},
    "unsigned": {"Cuint"},
    "double": {"Cdouble"},
    "long long": {"Clonglong"},
    "uint64_t": {"UInt64"},
    "int64_t": {"Int64"},
    "size_t": {"Csize_t"},
    "double*": {"Ptr{Float64}"},
    "int*": {"Ptr{Cint}"},
    "char*": {"Ptr{UInt8}", "Cstring"},
}
RET2J = {"int": "Cint", "const char*": "Cstring", "size_t": "Csize_t"}


def _norm(ctype):
    t = re.sub(r"\bconst\b", "", ctype)
    t = re.sub(r"\s+", " ", t).strip()
    t = t.replace(" *", "*")
    return t


def header_prototypes():
    src = open(os.path.join(ROOT, "include", "dfm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(?m)^\s*((?:const\s+)?[a-z_0-9 ]+?\*?)\s*\b(dfm_[a-z_0-9]+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        ret = re.sub(r"\s+", " ", ret).strip().replace(" *", "*")
        types = []
        if args.strip() and args.strip() != "void":
            for a in args.split(","):
                a = re.sub(r"/\*.*?\*/", "", a).strip()
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z_0-9]*)$", a)          # strip the parameter name
                types.append(_norm(mm.group(1)))
        protos[name] = (ret, types)
    return protos


def julia_ccalls():
    src = open(os.path.join(ROOT, "julia", "dfm_hip.jl")).read()
    src = re.sub(r"(?m)#.*$", "", src)                                        # comments
    calls = []
    for m in re.finditer(r"ccall\(\(:(dfm_[a-z_0-9]+),\s*LIB\),\s*([A-Za-z0-9{}]+),\s*\(", src):
        name, ret = m.group(1), m.group(2)
        i = m.end()
        depth, j = 1, i
        while depth:                                                          # the argument-type tuple, balanced
            c = src[j]
            depth += (c == "(") - (c == ")")
            j += 1
        tup = src[i:j - 1]
        parts, cur, d = [], "", 0
        for c in tup:
            if c == "{":
                d += 1
            elif c == "}":
                d -= 1
            if c == "," and d == 0:
                parts.append(cur.strip()); cur = ""
            else:
                cur += c
        if cur.strip():
            parts.append(cur.strip())
        calls.append((name, ret, parts))
    return calls


def test_header_parser_sees_every_prototype():
    names = set(re.findall(r"\b(dfm_[a-z_0-9]+)\s*\(", re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "dfm_hip.h")).read(), flags=re.S)))
    assert set(header_prototypes()) == names


def test_every_julia_ccall_matches_the_header():
    protos = header_prototypes()
    calls = julia_ccalls()
    assert len(calls) >= 15, "ccall parser found too few calls"
    for name, ret, jargs in calls:
        assert name in protos, f"{name}: not declared in include/dfm_hip.h"
        cret, cargs = protos[name]
        assert RET2J[cret] == ret, f"{name}: return type {ret} vs C {cret}"
        assert len(jargs) == len(cargs), f"{name}: {len(jargs)} ccall argument types vs {len(cargs)} C parameters"
        for k, (ja, ca) in enumerate(zip(jargs, cargs)):
            assert ca in C2J, f"{name}: unmapped C type {ca!r}"
            assert ja in C2J[ca], f"{name}: argument {k}: Julia {ja} vs C {ca}"


def test_the_shim_binds_the_batched_and_multi_gpu_entries():
    bound = {name for name, _, _ in julia_ccalls()}
    for need in ("dfm_create", "dfm_destroy", "dfm_last_error", "dfm_pca_init_batch", "dfm_em_batch", "dfm_em_varp_batch",
                 "dfm_ks_pass_batch", "dfm_ks_pass_batch_multi", "dfm_ks_pass_ar_batch", "dfm_als_batch",
                 "dfm_ols_batch", "dfm_chow_batch", "dfm_var_bootstrap_irf", "dfm_quantile_bands",
                 # the library's multi-GPU object (one communicator for the session, resident jobs, device generation)
                 "dfm_multi_create", "dfm_multi_destroy", "dfm_multi_last_error", "dfm_multi_load", "dfm_multi_synth",
                 "dfm_multi_ks_pass", "dfm_multi_em", "dfm_multi_fetch"):
        assert need in bound, need


def test_the_shim_batches_the_factor_number_tables():
    """estimate_factor_numbers / amengual_watson_test (dfm_functions.ipynb:698-768) reach the GPU as batched calls: the
    forwarding method packs every static run into one dfm_als_batch call and every dynamic run of every static count into
    a second one (r_each per run), and returns the reference's own FactorNumberEstimateStats."""
    src = open(os.path.join(ROOT, "julia", "dfm_hip.jl")).read()
    body = src[src.index("function estimate_factor_numbers_hip"):]
    body = body[:body.index("\nend\n")]
    assert body.count("DFMHip.als_batch(") == 2                               # static runs; all dynamic runs
    assert "FactorNumberEstimateStats(" in body and "lagmat(" in body and "initperiod+4" in body
    als = src[src.index("function als_batch"):]
    als = als[:als.index("\nend\n")]
    assert "r_each" in als and ":dfm_als_batch" in als and "shared ? 0 : T * N" in als
    em = src[src.index("function em_batch"):]
    em = em[:em.index("\nend\n")]
    assert "multi(ngpu)" in em and "multi_em(" in em                          # the cached object, not a communicator per call


def test_the_shim_never_drops_loading_constraints():
    src = open(os.path.join(ROOT, "julia", "dfm_hip.jl")).read()
    body = src[src.index("function estimate_factor_hip!"):]
    body = body[:body.index("\nend\n")]
    assert "lam_constr = nothing" in body and "lam_constr !== nothing" in body and "error(" in body
    est = src[src.index("function estimate!(m::DFMModel, ::Parametric"):]
    est = est[:est.index("\nend\n")]
    assert "nt_min_factor_estimation" in est                                  # the `enough` filter of api.estimate
    assert "lam_constr_f === nothing" in est and "nrep" in est and "ngpu" in est


def test_the_shim_repairs_the_scalar_shock_impulse_response():
    """dfm_functions.ipynb:817-821 calls the five-argument compute_irf_single_shock! with six arguments and an undefined `x`; the shim's
    replacement must call it with (irfs, varm, 1, shock_id, T) on a ny x T matrix."""
    body = src_of("function impulse_response(varm::VARModel, shock_id::Real, T::Integer)")
    assert "Matrix{Float64}(undef, size(varm.Q, 1), T)" in body
    assert "compute_irf_single_shock!(irfs, varm, 1, Int(shock_id), T)" in body


def src_of(signature):
    src = open(os.path.join(ROOT, "julia", "dfm_hip.jl")).read()
    start = src.index(signature)
    return src[start:src.index("\nend\n", start)]
