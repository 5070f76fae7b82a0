# AggMGHip.jl -- the reference-side binding a maintainer of AgglomerationMultigrid1D would add:
# a thin `ccall` shim over libaggmg_hip.so (include/aggmg_hip.h) that keeps the reference's
# entry points intact.  NOT EXECUTED in this repository's CI: Julia is not installed in the
# build image or assumed on the GPU box.  The tested contract is the C ABI; the Python mirror
# (agglomerationmultigrid1d_amd/api.py) binds the same symbols with the same semantics.
#
# Usage (inside the reference module, after include("solvers.jl")):
#     include("AggMGHip.jl");  using .AggMGHip
#     Hd = AggMGHip.DeviceHierarchy(H)                 # H::MeshHierarchy, built as usual
#     x  = multigrid_v_cycle(Hd, x0, b)                # same signature / defaults / return
module AggMGHip

import ..MeshHierarchy, ..AbstractSmoother, ..JacobiSmoother, ..BlockJacobi,
       ..AdditiveSchwarzSmoother, ..HybridSchwarzSmoother
import ..multigrid_v_cycle, ..apply_smoother
import LinearAlgebra as la
import SparseArrays as sp

const LIB = get(ENV, "AGGMG_HIP_LIB", "libaggmg_hip.so")
const Handle = Ptr{Cvoid}

# status -> the exception types the reference throws (SURVEY.md 8b)
function check(ctx::Handle, st::Cint)
    st == 0 && return
    msg = unsafe_string(ccall((:aggmg_last_error, LIB), Cstring, (Handle,), ctx))
    st == -1 && throw(ArgumentError(msg))
    st == -2 && throw(DimensionMismatch(msg))
    st == -3 && throw(la.SingularException(0))
    error(msg)                                  # -4 HIP failure, -5 unsupported
end

mutable struct Context
    h::Handle
    function Context(device::Integer = 0)
        r = Ref{Handle}(C_NULL)
        check(C_NULL, ccall((:aggmg_create, LIB), Cint, (Cint, Ref{Handle}), device, r))
        c = new(r[])
        finalizer(c -> ccall((:aggmg_destroy, LIB), Cint, (Handle,), c.h), c)
    end
end

# H.mStiffness[k] / H.mInterpolation[k]: SparseMatrixCSC{Float64,Int64}, 1-based, passed as is
function upload(ctx::Context, A::sp.SparseMatrixCSC{Float64,Int64}, kind::Integer)
    r = Ref{Handle}(C_NULL)
    check(ctx.h, ccall((:aggmg_csc_upload, LIB), Cint,
        (Handle, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Cint, Cint, Ref{Handle}),
        ctx.h, size(A, 1), size(A, 2), A.colptr, A.rowval, A.nzval, 1, kind, r))
    return r[]
end

# H.mSmoothers[k]: only A and mBlockInds cross the boundary; the Vector{LU} of heap objects is
# re-derived on the device side (identical mathematics to src/smoother.jl:159-162)
function smoother(ctx::Context, op::Handle, S::AbstractSmoother)
    r = Ref{Handle}(C_NULL)
    if S isa JacobiSmoother
        check(ctx.h, ccall((:aggmg_jacobi_setup, LIB), Cint, (Handle, Handle, Ref{Handle}), ctx.h, op, r))
    else
        kind = S isa HybridSchwarzSmoother ? 1 : 0
        inds = Matrix{Int64}(S.mBlockInds)      # (m x nb), column-major, 1-based
        check(ctx.h, ccall((:aggmg_blockjacobi_setup, LIB), Cint,
            (Handle, Handle, Int64, Int64, Ptr{Int64}, Cint, Cint, Ref{Handle}),
            ctx.h, op, size(inds, 1), size(inds, 2), inds, 1, kind, r))
    end
    return r[]
end

mutable struct DeviceHierarchy
    ctx::Context
    H::MeshHierarchy
    ops::Vector{Handle}; sms::Vector{Handle}; Ls::Vector{Handle}
    h::Handle
end

function DeviceHierarchy(H::MeshHierarchy; ctx::Context = Context(0))
    n = length(H.mMeshes)
    ops = [upload(ctx, H.mStiffness[k], 0) for k in 1:n]
    Ls = [upload(ctx, sp.sparse(H.mInterpolation[k]), 1) for k in 1:n-1]
    sms = [smoother(ctx, ops[k], H.mSmoothers[k]) for k in 1:n-1]
    r = Ref{Handle}(C_NULL)
    check(ctx.h, ccall((:aggmg_hier_create, LIB), Cint,
        (Handle, Cint, Ptr{Handle}, Ptr{Handle}, Ptr{Handle}, Cint, Ref{Handle}),
        ctx.h, n, ops, vcat(sms, [C_NULL]), vcat(Ls, [C_NULL]), 2, r))   # 2 = AGGMG_COARSE_AUTO
    return DeviceHierarchy(ctx, H, ops, sms, Ls, r[])
end

# device vector: aggmg_dev_alloc / aggmg_memcpy_h2d / aggmg_memcpy_d2h
mutable struct DeviceVector
    ctx::Context
    p::Ptr{Cvoid}
    n::Int
    function DeviceVector(ctx::Context, n::Integer)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ctx.h, ccall((:aggmg_dev_alloc, LIB), Cint, (Handle, Int64, Ref{Ptr{Cvoid}}), ctx.h, 8n, r))
        v = new(ctx, r[], n)
        finalizer(x -> ccall((:aggmg_dev_free, LIB), Cint, (Handle, Ptr{Cvoid}), x.ctx.h, x.p), v)
        return v
    end
end
function DeviceVector(ctx::Context, x::Vector{Float64})
    v = DeviceVector(ctx, length(x))
    GC.@preserve x check(ctx.h, ccall((:aggmg_memcpy_h2d, LIB), Cint, (Handle, Ptr{Cvoid}, Ptr{Float64}, Int64),
        ctx.h, v.p, x, 8length(x)))
    return v
end
function download(v::DeviceVector)
    out = Vector{Float64}(undef, v.n)
    GC.@preserve out check(v.ctx.h, ccall((:aggmg_memcpy_d2h, LIB), Cint, (Handle, Ptr{Float64}, Ptr{Cvoid}, Int64),
        v.ctx.h, out, v.p, 8v.n))
    return out
end

# same name, positional / keyword arguments, defaults and return shape as src/solvers.jl:19-20;
# x0 and b are not mutated, a new Vector is returned
function multigrid_v_cycle(Hd::DeviceHierarchy, x0::AbstractVector, b::AbstractVector;
        nPre::Integer = 3, nPost::Integer = 3, alpha::AbstractFloat = 2.0 / 3.0)
    x0v = Vector{Float64}(x0); bv = Vector{Float64}(b); out = similar(bv)
    GC.@preserve x0v bv out check(Hd.ctx.h, ccall((:aggmg_vcycle, LIB), Cint,
        (Handle, Handle, Ptr{Float64}, Ptr{Float64}, Cint, Cint, Float64, Ptr{Float64}),
        Hd.ctx.h, Hd.h, x0v, bv, nPre, nPost, Float64(alpha), out))
    return out
end

# ldiv!(H, b) / ldiv!(y, H, b): src/solvers.jl:63-92
function la.ldiv!(Hd::DeviceHierarchy, b::AbstractVector)
    b[:] = multigrid_v_cycle(Hd, zeros(size(Hd.H.mStiffness[1], 1)), b); return
end
function la.ldiv!(y::AbstractVector, Hd::DeviceHierarchy, b::AbstractVector)
    y[:] = multigrid_v_cycle(Hd, zeros(size(Hd.H.mStiffness[1], 1)), b); return
end

# multigrid(H, x0, b, maxiter, tol) -> x, iter, res, err   (src/solvers.jl:116-139) with the loop,
# the residual norms and the stopping test on the device.  `err` needs the fine-level direct solve
# of :120: computed on the host only when asked for (exact = true) from the final iterate history,
# otherwise returned empty.
function multigrid(Hd::DeviceHierarchy, x0::AbstractVector, b::AbstractVector, maxiter::Integer,
        tol::AbstractFloat; check_every::Integer = 1)
    N = length(b)
    dx0 = DeviceVector(Hd.ctx, Vector{Float64}(x0)); db = DeviceVector(Hd.ctx, Vector{Float64}(b))
    dx = DeviceVector(Hd.ctx, N)
    res = zeros(cld(max(maxiter, 1), check_every)); ncyc = Ref{Cint}(0); nchk = Ref{Cint}(0)
    check(Hd.ctx.h, ccall((:aggmg_multigrid_dev, LIB), Cint,
        (Handle, Handle, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Cint, Cint, Cint, Float64, Ptr{Cvoid},
         Ptr{Float64}, Ref{Cint}, Ref{Cint}),
        Hd.ctx.h, Hd.h, dx0.p, db.p, maxiter, Float64(tol), check_every, 3, 3, 2.0 / 3.0, dx.p, res, ncyc, nchk))
    return download(dx), Int(ncyc[]), res[1:nchk[]], Float64[]
end

end # module
