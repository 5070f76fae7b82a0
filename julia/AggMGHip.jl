# AggMGHip.jl -- the reference-side binding a maintainer of AgglomerationMultigrid1D would add:
# a thin `ccall` shim over libaggmg_hip.so (include/aggmg_hip.h) that keeps the reference's
# entry points intact.  NOT EXECUTED in this repository's CI: Julia is not installed in the
# build image or assumed on the GPU box (written and reviewed by inspection only).  The tested
# contract is the C ABI; the Python mirror (agglomerationmultigrid1d_amd/api.py) binds the same
# symbols with the same semantics.
#
# Usage (inside the reference module, after include("solvers.jl")):
#     include("AggMGHip.jl")
#     Hd = AggMGHip.DeviceHierarchy(H)                 # H::MeshHierarchy, built as usual
#     x  = multigrid_v_cycle(Hd, x0, b)                # same signature / defaults / return
#     x, iter, res, err = multigrid(Hd, x0, b, 100, 1e-10)          # device loop, res AND err histories formed on the device
#     xd = multigrid_v_cycle(Hd, DeviceVector(Hd.ctx, x0), DeviceVector(Hd.ctx, b))   # vectors stay in HBM
# Host arrays cross PCIe on every call (measured: 17 ms per config-3 cycle that computes in 0.7 ms); callers
# that loop hand over DeviceVectors; `multigrid` / `iterative_smoother_solve` always loop on the device.
#
# Every function of the reference that this file gives a device method is IMPORTED and EXTENDED
# (a `function f(...)` on an imported name adds a method; on a non-imported name it would define
# a new, unrelated AggMGHip.f):
#     multigrid_v_cycle, multigrid, iterative_smoother_solve   src/solvers.jl:19,116,189
#     apply_smoother, dg_smoother, cg_smoother                 src/smoother.jl:6,88,142
#     LinearAlgebra.ldiv!                                      src/solvers.jl:63,84
# Device objects plug into the reference's own seams (SURVEY.md 8b):
#     DeviceSmoother <: AbstractSmoother                       src/AgglomerationMultigrid1D.jl:16
#     DeviceTransfer <: AbstractMatrix{Float64}                mInterpolation, src/mesh_heirarchy.jl:26
module AggMGHip

import ..MeshHierarchy, ..AbstractSmoother, ..JacobiSmoother, ..BlockJacobi,
       ..AdditiveSchwarzSmoother, ..HybridSchwarzSmoother, ..CgMesh
import ..multigrid_v_cycle, ..multigrid, ..iterative_smoother_solve
import ..apply_smoother, ..dg_smoother, ..cg_smoother
import ..dg_dg_interpolation, ..aggdg_dg_interpolation, ..aggdg_aggdg_interpolation
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

# ---------------------------------------------------------------------------------------------
# context and device objects.  Lifetime: every device object holds a strong reference to its
# Context, so the context outlives them whatever order the GC finalizes in; a finalizer that
# runs after close(ctx) finds `ctx.h == C_NULL` and does nothing (aggmg_destroy released it all).
# ---------------------------------------------------------------------------------------------
mutable struct Context
    h::Handle
    function Context(device::Integer = 0)
        r = Ref{Handle}(C_NULL)
        check(C_NULL, ccall((:aggmg_create, LIB), Cint, (Cint, Ref{Handle}), device, r))
        c = new(r[])
        finalizer(close, c)
        return c
    end
end
function Base.close(c::Context)
    if c.h != C_NULL
        ccall((:aggmg_destroy, LIB), Cint, (Handle,), c.h)
        c.h = C_NULL
    end
    return nothing
end

const DEFAULT_CONTEXT = Ref{Union{Nothing,Context}}(nothing)
function default_context()
    DEFAULT_CONTEXT[] === nothing && (DEFAULT_CONTEXT[] = Context(0))
    return DEFAULT_CONTEXT[]
end

# device mirror of one SparseMatrixCSC{Float64,Int64} (H.mStiffness[k] / H.mInterpolation[k]):
# Julia's colptr / rowval / nzval are passed as they are (1-based)
mutable struct DeviceOperator
    ctx::Context
    h::Handle
    m::Int
    n::Int
    function DeviceOperator(ctx::Context, A::sp.SparseMatrixCSC{Float64,Int64}, kind::Integer)
        r = Ref{Handle}(C_NULL)
        GC.@preserve A check(ctx.h, ccall((:aggmg_csc_upload, LIB), Cint,
            (Handle, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Cint, Cint, Ref{Handle}),
            ctx.h, size(A, 1), size(A, 2), A.colptr, A.rowval, A.nzval, 1, kind, r))
        op = new(ctx, r[], size(A, 1), size(A, 2))
        finalizer(free!, op)
        return op
    end
end
function free!(op::DeviceOperator)
    if op.h != C_NULL && op.ctx.h != C_NULL
        ccall((:aggmg_op_free, LIB), Cint, (Handle, Handle), op.ctx.h, op.h)
    end
    op.h = C_NULL
    return nothing
end

# ---- smoother seam: a device-resident AbstractSmoother ----------------------------------------
mutable struct DeviceSmoother <: AbstractSmoother
    ctx::Context
    A::DeviceOperator          # keeps the operator alive (the C handle refers to it)
    h::Handle
end
function free!(S::DeviceSmoother)
    if S.h != C_NULL && S.ctx.h != C_NULL
        ccall((:aggmg_smoother_free, LIB), Cint, (Handle, Handle), S.ctx.h, S.h)
    end
    S.h = C_NULL
    return nothing
end
function _wrap_smoother(ctx::Context, op::DeviceOperator, h::Handle)
    S = DeviceSmoother(ctx, op, h)
    finalizer(free!, S)
    return S
end

element_nodes(mesh) = Matrix{Int64}(reduce(hcat, [el.mNodesInd for el in mesh.mElements]))   # (p+1) x n, 1-based

# JacobiSmoother(Diagonal(A)); with a CgMesh at hand the element node lists go along so that the level
# runs the fused chain kernel (aggmg_jacobi_setup_elements) -- same smoother either way
function device_jacobi(op::DeviceOperator, mesh = nothing)
    r = Ref{Handle}(C_NULL)
    if mesh isa CgMesh
        inds = element_nodes(mesh)
        GC.@preserve inds check(op.ctx.h, ccall((:aggmg_jacobi_setup_elements, LIB), Cint,
            (Handle, Handle, Int64, Int64, Ptr{Int64}, Cint, Ref{Handle}),
            op.ctx.h, op.h, size(inds, 1), size(inds, 2), inds, 1, r))
    else
        check(op.ctx.h, ccall((:aggmg_jacobi_setup, LIB), Cint, (Handle, Handle, Ref{Handle}), op.ctx.h, op.h, r))
    end
    return _wrap_smoother(op.ctx, op, r[])
end

# BlockJacobi / AdditiveSchwarz (kind 0), HybridSchwarz (kind 1): only A and mBlockInds cross the
# boundary; the Vector{LU} of heap objects is re-derived on the device side (identical mathematics
# to src/smoother.jl:159-162).  kind 2: red-black block Gauss-Seidel (extension, SURVEY D1).
function device_block_smoother(op::DeviceOperator, mBlockInds::AbstractMatrix{<:Integer}, kind::Integer)
    inds = Matrix{Int64}(mBlockInds)          # (m x nb), column-major, 1-based
    r = Ref{Handle}(C_NULL)
    GC.@preserve inds check(op.ctx.h, ccall((:aggmg_blockjacobi_setup, LIB), Cint,
        (Handle, Handle, Int64, Int64, Ptr{Int64}, Cint, Cint, Ref{Handle}),
        op.ctx.h, op.h, size(inds, 1), size(inds, 2), inds, 1, kind, r))
    return _wrap_smoother(op.ctx, op, r[])
end

# the reference's smoother objects -> device smoothers (used by DeviceHierarchy(H))
function device_smoother(op::DeviceOperator, S::AbstractSmoother, mesh = nothing)
    S isa DeviceSmoother && return S
    S isa JacobiSmoother && return device_jacobi(op, mesh)
    return device_block_smoother(op, S.mBlockInds, S isa HybridSchwarzSmoother ? 1 : 0)
end

# dg_smoother(dgMesh, A, :jac | :blockJac)   src/smoother.jl:142-168 -- device variants: the reference's
# positional signature with the operator already on the device (dispatch on DeviceOperator, so the
# reference's own methods stay untouched and selectable: `backend = :cpu` is simply calling them)
function dg_smoother(dgMesh, A::DeviceOperator, smootherType::Symbol)
    smootherType == :jac && return device_jacobi(A)
    smootherType == :blockJac && return device_block_smoother(A, element_nodes(dgMesh), 0)
    smootherType == :blockGS && return device_block_smoother(A, element_nodes(dgMesh), 2)   # extension
    throw(ArgumentError("dg_smoother: unknown smoother type $smootherType"))
end
# cg_smoother(cgMesh, A, :jac | :addSchwarz | :hybridSchwarz)   src/smoother.jl:88-139
function cg_smoother(cgMesh, A::DeviceOperator, smootherType::Symbol)
    smootherType == :jac && return device_jacobi(A, cgMesh)
    smootherType == :addSchwarz && return device_block_smoother(A, element_nodes(cgMesh), 0)
    smootherType == :hybridSchwarz && return device_block_smoother(A, element_nodes(cgMesh), 1)
    throw(ArgumentError("cg_smoother: unknown smoother type $smootherType"))
end
# ... and with a backend switch on the reference's own signature (SURVEY.md 7 step 2): backend = :cpu
# returns the reference's smoother, :hip uploads A and returns a DeviceSmoother
function dg_smoother(dgMesh, A::sp.SparseMatrixCSC{Float64,Int64}, smootherType::Symbol, backend::Symbol;
        ctx::Context = default_context())
    backend == :cpu && return dg_smoother(dgMesh, A, smootherType)
    return dg_smoother(dgMesh, DeviceOperator(ctx, A, 0), smootherType)
end
function cg_smoother(cgMesh, A::sp.SparseMatrixCSC{Float64,Int64}, smootherType::Symbol, backend::Symbol;
        ctx::Context = default_context())
    backend == :cpu && return cg_smoother(cgMesh, A, smootherType)
    return cg_smoother(cgMesh, DeviceOperator(ctx, A, 0), smootherType)
end

# apply_smoother(A::AbstractSmoother, B::AbstractVecOrMat; alpha = 1.0) -> new array   src/smoother.jl:6,30,56,69
function apply_smoother(S::DeviceSmoother, B::AbstractVecOrMat; alpha::Float64 = 1.0)
    Bm = Matrix{Float64}(reshape(B, size(B, 1), :))          # column-major N x ncols
    Y = similar(Bm)
    GC.@preserve Bm Y check(S.ctx.h, ccall((:aggmg_smoother_apply, LIB), Cint,
        (Handle, Handle, Ptr{Float64}, Int64, Int64, Float64, Ptr{Float64}),
        S.ctx.h, S.h, Bm, size(Bm, 1), size(Bm, 2), alpha, Y))
    return B isa AbstractVector ? vec(Y) : Y
end

# ---- transfer seam: a device-resident AbstractMatrix{Float64} used as L*v and L'*v ---------------
struct DeviceTransfer <: AbstractMatrix{Float64}
    op::DeviceOperator
    host::sp.SparseMatrixCSC{Float64,Int64}     # kept for getindex / display; never used by * or '
end
DeviceTransfer(ctx::Context, L::AbstractMatrix{Float64}) = (Ls = sp.sparse(L); DeviceTransfer(DeviceOperator(ctx, Ls, 1), Ls))
Base.size(L::DeviceTransfer) = (L.op.m, L.op.n)
Base.getindex(L::DeviceTransfer, i::Int, j::Int) = L.host[i, j]
# L * uc   (src/solvers.jl:42 computes u + L*uc: prolong_add on a zero vector gives L*uc)
function Base.:*(L::DeviceTransfer, v::AbstractVector)
    vv = Vector{Float64}(v); out = zeros(L.op.m)
    GC.@preserve vv out check(L.op.ctx.h, ccall((:aggmg_prolong_add, LIB), Cint,
        (Handle, Handle, Ptr{Float64}, Ptr{Float64}), L.op.ctx.h, L.op.h, vv, out))
    return out
end
# L' * r   (src/solvers.jl:36); `L'` is the lazy la.Adjoint wrapper AbstractMatrix provides
function Base.:*(Lt::la.Adjoint{Float64,DeviceTransfer}, r::AbstractVector)
    L = parent(Lt)
    rv = Vector{Float64}(r); out = Vector{Float64}(undef, L.op.n)
    GC.@preserve rv out check(L.op.ctx.h, ccall((:aggmg_restrict, LIB), Cint,
        (Handle, Handle, Ptr{Float64}, Ptr{Float64}), L.op.ctx.h, L.op.h, rv, out))
    return out
end

# ---- hierarchy -----------------------------------------------------------------------------------
mutable struct DeviceHierarchy
    ctx::Context
    H::MeshHierarchy
    ops::Vector{DeviceOperator}
    sms::Vector{DeviceSmoother}
    Ls::Vector{DeviceTransfer}
    h::Handle
end

# ONE finalizer frees in dependency order: hierarchy, then smoothers, then operators; the context is
# only referenced (its own finalizer cannot run before this object is unreachable)
function free!(Hd::DeviceHierarchy)
    if Hd.h != C_NULL && Hd.ctx.h != C_NULL
        ccall((:aggmg_hier_free, LIB), Cint, (Handle, Handle), Hd.ctx.h, Hd.h)
    end
    Hd.h = C_NULL
    foreach(free!, Hd.sms)
    foreach(L -> free!(L.op), Hd.Ls)
    foreach(free!, Hd.ops)
    return nothing
end

# coarse_mode 2 = AGGMG_COARSE_AUTO
function DeviceHierarchy(H::MeshHierarchy; ctx::Context = default_context(), coarse_mode::Integer = 2)
    n = length(H.mMeshes)
    ops = [DeviceOperator(ctx, H.mStiffness[k], 0) for k in 1:n]
    Ls = [H.mInterpolation[k] isa DeviceTransfer ? H.mInterpolation[k] : DeviceTransfer(ctx, H.mInterpolation[k])
          for k in 1:n-1]
    sms = DeviceSmoother[device_smoother(ops[k], H.mSmoothers[k], H.mMeshes[k]) for k in 1:n-1]
    r = Ref{Handle}(C_NULL)
    oph = Handle[o.h for o in ops]
    smh = vcat(Handle[s.h for s in sms], [C_NULL])
    lh = vcat(Handle[L.op.h for L in Ls], [C_NULL])
    GC.@preserve oph smh lh check(ctx.h, ccall((:aggmg_hier_create, LIB), Cint,
        (Handle, Cint, Ptr{Handle}, Ptr{Handle}, Ptr{Handle}, Cint, Ref{Handle}),
        ctx.h, n, oph, smh, lh, coarse_mode, r))
    Hd = DeviceHierarchy(ctx, H, ops, sms, Ls, r[])
    finalizer(free!, Hd)
    return Hd
end

# MeshHierarchy(mMeshes, mBdConds, A, G, D, C; nDG, nAgg) with the agglomerated levels the reference's
# DG-fine constructor sizes but never fills (src/mesh_heirarchy.jl:140-181; SURVEY D4).  LABELLED
# EXTENSION: the loop below is the recurrence of the CG-fine constructor's agglomerated branch
# (src/mesh_heirarchy.jl:89-106) applied after the DG levels; it calls the struct's positional
# constructor (:136-137).  This is the hierarchy shape of BASELINE configs 3 / 4.
function dg_agg_hierarchy(mMeshes, mBdConds, A, G, D, C; nDG::Integer = 1, nAgg::Integer = 0)
    nDG <= 0 && throw(ArgumentError("At least one DG mesh required."))
    length(mMeshes) != nDG + nAgg && throw(ArgumentError(
        "Length of vector of meshes does not match inputed number of DG and agglomerated meshes."))
    n = nDG + nAgg
    St = Vector{sp.SparseMatrixCSC{Float64,Int64}}(undef, n)
    Gs = Vector{sp.SparseMatrixCSC{Float64,Int64}}(undef, n)
    Ds = Vector{sp.SparseMatrixCSC{Float64,Int64}}(undef, n)
    Cs = Vector{sp.SparseMatrixCSC{Float64,Int64}}(undef, n)
    Sm = Vector{AbstractSmoother}(undef, n)
    Li = Vector{AbstractMatrix{Float64}}(undef, n - 1)
    Gs[1], Ds[1], Cs[1], St[1] = G, D, C, A
    Sm[1] = dg_smoother(mMeshes[1], A, :blockJac)
    for i in 1:(n-1)
        L = i <= nDG - 1 ? dg_dg_interpolation(mMeshes[i+1], mMeshes[i]) :
            i == nDG     ? aggdg_dg_interpolation(mMeshes[i+1], mMeshes[i]) :
                           aggdg_aggdg_interpolation(mMeshes[i+1], mMeshes[i], mMeshes[nDG])
        Li[i] = L
        Gs[i+1] = L' * Gs[i] * L
        Ds[i+1] = L' * Ds[i] * L
        Cs[i+1] = L' * Cs[i] * L
        St[i+1] = Cs[i+1] - Ds[i+1] * (mMeshes[i+1].mMassMatrixLU \ Gs[i+1])
        Sm[i+1] = dg_smoother(mMeshes[i+1], St[i+1], :blockJac)
    end
    return MeshHierarchy(mMeshes, St, Gs, Ds, Cs, Sm, Li, mBdConds)
end

# device vector: aggmg_dev_alloc (zeroed) / aggmg_memcpy_h2d / aggmg_memcpy_d2h.  What a caller keeps between
# calls so that nothing but the handles crosses PCIe: multigrid_v_cycle, ldiv! and multigrid take and return them.
# LIFETIME RULE (the use-after-free of round 3, api.DeviceVector.ptr in the Python mirror): `v.p` is a bare address --
# the GC may finalise `v` (aggmg_dev_free) as soon as no Julia reference to it is live, even while a ccall that was
# handed `v.p` is still being set up.  Every ccall that passes `v.p` therefore stands under `GC.@preserve v`
# (tests/test_julia_shim_lint.py checks it); once the call has enqueued its launches the free is safe, because
# aggmg_dev_free synchronises the context's stream first.
mutable struct DeviceVector
    ctx::Context               # strong reference: the context outlives the vector
    p::Ptr{Cvoid}
    n::Int
    function DeviceVector(ctx::Context, n::Integer)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ctx.h, ccall((:aggmg_dev_alloc, LIB), Cint, (Handle, Int64, Ref{Ptr{Cvoid}}), ctx.h, 8n, r))
        v = new(ctx, r[], n)
        finalizer(free!, v)
        return v
    end
end
function free!(v::DeviceVector)
    if v.p != C_NULL && v.ctx.h != C_NULL     # after close(ctx) the memory is gone with the context
        ccall((:aggmg_dev_free, LIB), Cint, (Handle, Ptr{Cvoid}), v.ctx.h, v.p)
    end
    v.p = C_NULL
    return nothing
end
function DeviceVector(ctx::Context, x::AbstractVector)
    x = Vector{Float64}(x)
    v = DeviceVector(ctx, length(x))
    GC.@preserve v x check(ctx.h, ccall((:aggmg_memcpy_h2d, LIB), Cint, (Handle, Ptr{Cvoid}, Ptr{Float64}, Int64),
        ctx.h, v.p, x, 8length(x)))
    return v
end
Base.length(v::DeviceVector) = v.n
Base.size(v::DeviceVector) = (v.n,)
function download(v::DeviceVector)
    out = Vector{Float64}(undef, v.n)
    GC.@preserve v out check(v.ctx.h, ccall((:aggmg_memcpy_d2h, LIB), Cint, (Handle, Ptr{Float64}, Ptr{Cvoid}, Int64),
        v.ctx.h, out, v.p, 8v.n))
    return out
end
Base.Vector(v::DeviceVector) = download(v)
Base.Array(v::DeviceVector) = download(v)

# ---- solvers (src/solvers.jl) --------------------------------------------------------------------
# same name, positional / keyword arguments, defaults and return shape as src/solvers.jl:19-20;
# x0 and b are not mutated, a new Vector is returned
function multigrid_v_cycle(Hd::DeviceHierarchy, x0::AbstractVector, b::AbstractVector;
        nPre::Integer = 3, nPost::Integer = 3, alpha::AbstractFloat = 2.0 / 3.0)
    # (convert: no copy when the caller's vectors are Vector{Float64} already -- arrays page-locked with pin! stay the ones
    # the library reads)
    x0v = convert(Vector{Float64}, x0); bv = convert(Vector{Float64}, b); out = similar(bv)
    GC.@preserve x0v bv out check(Hd.ctx.h, ccall((:aggmg_vcycle, LIB), Cint,
        (Handle, Handle, Ptr{Float64}, Ptr{Float64}, Cint, Cint, Float64, Ptr{Float64}),
        Hd.ctx.h, Hd.h, x0v, bv, nPre, nPost, Float64(alpha), out))
    return out
end

# the same cycle on vectors that are already in HBM: nothing is copied, a new DeviceVector is returned and
# x0 / b are left as they are (aggmg_vcycle_dev) -- the form for callers that loop
function multigrid_v_cycle(Hd::DeviceHierarchy, x0::DeviceVector, b::DeviceVector;
        nPre::Integer = 3, nPost::Integer = 3, alpha::AbstractFloat = 2.0 / 3.0)
    (x0.n == b.n) || throw(DimensionMismatch("multigrid_v_cycle: x0 and b differ in length"))
    out = DeviceVector(Hd.ctx, b.n)
    GC.@preserve x0 b out check(Hd.ctx.h, ccall((:aggmg_vcycle_dev, LIB), Cint,
        (Handle, Handle, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Float64, Ptr{Cvoid}),
        Hd.ctx.h, Hd.h, x0.p, b.p, nPre, nPost, Float64(alpha), out.p))
    return out
end

# ldiv!(y, H, b) with y, b in HBM: one V-cycle from a zero guess written into y (y must not be b: the entry point
# refuses aliased output; ldiv!(H, b) for a DeviceVector goes through a fresh vector and swaps the storage)
function la.ldiv!(y::DeviceVector, Hd::DeviceHierarchy, b::DeviceVector)
    GC.@preserve b y check(Hd.ctx.h, ccall((:aggmg_vcycle_dev, LIB), Cint,      # x0 = C_NULL: zero initial guess
        (Handle, Handle, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Float64, Ptr{Cvoid}),
        Hd.ctx.h, Hd.h, C_NULL, b.p, 3, 3, 2.0 / 3.0, y.p))
    return
end
function la.ldiv!(Hd::DeviceHierarchy, b::DeviceVector)
    y = DeviceVector(Hd.ctx, b.n)
    la.ldiv!(y, Hd, b)
    b.p, y.p = y.p, b.p                                 # b now holds the result; y's finalizer frees the old storage
    return
end

# ldiv!(H, b) (overwrites b) / ldiv!(y, H, b): src/solvers.jl:63-92 -- one V-cycle from a zero guess
function la.ldiv!(Hd::DeviceHierarchy, b::AbstractVector)
    b[:] = multigrid_v_cycle(Hd, zeros(size(Hd.H.mStiffness[1], 1)), b); return
end
function la.ldiv!(y::AbstractVector, Hd::DeviceHierarchy, b::AbstractVector)
    y[:] = multigrid_v_cycle(Hd, zeros(size(Hd.H.mStiffness[1], 1)), b); return
end
# ldiv!(y, H, b) on Vector{Float64}s straight into y (no temporary result): with y and b page-locked once (pin!) -- the
# vectors of a Krylov loop that calls ldiv! every iteration -- the call moves them by DMA instead of staging them
function la.ldiv!(y::Vector{Float64}, Hd::DeviceHierarchy, b::Vector{Float64})
    (length(y) == length(b)) || throw(DimensionMismatch("ldiv!: y and b differ in length"))
    # x0 = C_NULL: the zero initial guess of ldiv! (include/aggmg_hip.h) -- no vector of zeros is sent
    GC.@preserve b y check(Hd.ctx.h, ccall((:aggmg_vcycle, LIB), Cint,
        (Handle, Handle, Ptr{Float64}, Ptr{Float64}, Cint, Cint, Float64, Ptr{Float64}),
        Hd.ctx.h, Hd.h, Ptr{Float64}(C_NULL), b, 3, 3, 2.0 / 3.0, y))
    return
end

# Page-locked host vectors (include/aggmg_hip.h: aggmg_host_register).  pin!(ctx, v) page-locks v's memory for good;
# v must stay alive and must not be resized until unpin!(ctx, v) (the context keeps a reference).
const PINNED = IdDict{Any,Any}()
function pin!(ctx::Context, v::Vector{Float64})
    GC.@preserve v check(ctx.h, ccall((:aggmg_host_register, LIB), Cint, (Handle, Ptr{Cvoid}, Int64),
        ctx.h, pointer(v), Int64(sizeof(v))))
    PINNED[v] = ctx
    return v
end
function unpin!(ctx::Context, v::Vector{Float64})
    GC.@preserve v check(ctx.h, ccall((:aggmg_host_unregister, LIB), Cint, (Handle, Ptr{Cvoid}), ctx.h, pointer(v)))
    delete!(PINNED, v)
    return v
end

# `A \\ b` of a device operator -- the fine-level direct solve behind the reference's `err` histories
# (u_exact = H.mStiffness[1] \\ b, src/solvers.jl:120; uExact = A \\ b, :194).  A one-level hierarchy of the operator IS
# the direct solve (:39): block cyclic reduction on the device when the operator is block-tridiagonal (every DG /
# agglomerated operator), the library's host banded LU otherwise (coarse_mode 2 = AGGMG_COARSE_AUTO).  Operators that
# neither takes (status -5: CG operators in the vertices-first numbering have no band) are solved on the host with the
# reference's own `\\` and uploaded -- once per call, never per cycle.
mutable struct DirectSolver
    ctx::Context
    op::DeviceOperator
    h::Handle                                  # one-level hierarchy, or C_NULL: host fallback
end
function free!(ds::DirectSolver)
    if ds.h != C_NULL && ds.ctx.h != C_NULL
        ccall((:aggmg_hier_free, LIB), Cint, (Handle, Handle), ds.ctx.h, ds.h)
    end
    ds.h = C_NULL
    return nothing
end
function DirectSolver(op::DeviceOperator)
    r = Ref{Handle}(C_NULL)
    oph = Handle[op.h]
    st = GC.@preserve oph ccall((:aggmg_hier_create, LIB), Cint,
        (Handle, Cint, Ptr{Handle}, Ptr{Handle}, Ptr{Handle}, Cint, Ref{Handle}),
        op.ctx.h, 1, oph, C_NULL, C_NULL, 2, r)
    st == -5 || check(op.ctx.h, st)             # -5: neither block-tridiagonal nor banded -> host `\\`
    ds = DirectSolver(op.ctx, op, st == 0 ? r[] : C_NULL)
    finalizer(free!, ds)
    return ds
end
# u = A \\ b as a DeviceVector; A_host: the reference's SparseMatrixCSC of the same operator (host fallback only)
function solve(ds::DirectSolver, A_host, b::DeviceVector)
    ds.h == C_NULL && return DeviceVector(ds.ctx, A_host \\ download(b))
    u = DeviceVector(ds.ctx, b.n); z = DeviceVector(ds.ctx, b.n)      # z: zero guess (aggmg_dev_alloc zeroes)
    GC.@preserve z b u check(ds.ctx.h, ccall((:aggmg_vcycle_dev, LIB), Cint,
        (Handle, Handle, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Float64, Ptr{Cvoid}),
        ds.ctx.h, ds.h, z.p, b.p, 0, 0, 1.0, u.p))
    return u
end
const DIRECT_SOLVERS = IdDict{Any,DirectSolver}()        # factored once per hierarchy / smoother, reused by every call
direct_solver(owner, op::DeviceOperator) = get!(() -> DirectSolver(op), DIRECT_SOLVERS, owner)

# multigrid(H, x0, b, maxiter, tol) -> x, iter, res, err   (src/solvers.jl:116-139): a METHOD OF THE
# REFERENCE'S FUNCTION for DeviceHierarchy, returning the reference's full 4-tuple.  The loop, the residual norms, the
# stopping test (:131) AND the error history err[i] = ||x_i - u_exact||_2 (:128) run on the device
# (aggmg_multigrid_dev): x0 and b go up once, x comes back once; u_exact = H.mStiffness[1] \\ b (:120) is solved once,
# on the device where the fine operator is block-tridiagonal (DirectSolver).  exact = false skips the direct solve
# and returns `err` empty.  nPre / nPost / alpha: the defaults of multigrid_v_cycle, as the reference's loop uses.
# x0, b may be DeviceVectors, in which case x is one too and nothing but the two histories leaves the device.
function multigrid(Hd::DeviceHierarchy, x0::Union{AbstractVector,DeviceVector}, b::Union{AbstractVector,DeviceVector},
        maxiter::Integer, tol::AbstractFloat; nPre::Integer = 3, nPost::Integer = 3,
        alpha::AbstractFloat = 2.0 / 3.0, exact::Bool = true, check_every::Integer = 1)
    on_device = x0 isa DeviceVector && b isa DeviceVector
    dx0 = x0 isa DeviceVector ? x0 : DeviceVector(Hd.ctx, x0)
    db = b isa DeviceVector ? b : DeviceVector(Hd.ctx, b)
    dx = DeviceVector(Hd.ctx, db.n)
    nchk = cld(max(maxiter, 1), check_every)
    res = zeros(nchk); err = zeros(nchk); ncyc = Ref{Cint}(0); nck = Ref{Cint}(0)
    if exact
        ue = solve(direct_solver(Hd, Hd.ops[1]), Hd.H.mStiffness[1], db)
        GC.@preserve dx0 db dx ue res err check(Hd.ctx.h, ccall((:aggmg_multigrid_dev, LIB), Cint,
            (Handle, Handle, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Cint, Cint, Cint, Float64, Ptr{Cvoid},
             Ptr{Float64}, Ref{Cint}, Ref{Cint}, Ptr{Cvoid}, Ptr{Float64}),
            Hd.ctx.h, Hd.h, dx0.p, db.p, maxiter, Float64(tol), check_every, nPre, nPost, Float64(alpha), dx.p,
            res, ncyc, nck, ue.p, err))
    else
        GC.@preserve dx0 db dx res check(Hd.ctx.h, ccall((:aggmg_multigrid_dev, LIB), Cint,
            (Handle, Handle, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Cint, Cint, Cint, Float64, Ptr{Cvoid},
             Ptr{Float64}, Ref{Cint}, Ref{Cint}, Ptr{Cvoid}, Ptr{Float64}),
            Hd.ctx.h, Hd.h, dx0.p, db.p, maxiter, Float64(tol), check_every, nPre, nPost, Float64(alpha), dx.p,
            res, ncyc, nck, C_NULL, C_NULL))
    end
    return (on_device ? dx : download(dx)), Int(ncyc[]), res[1:nck[]], (exact ? err[1:nck[]] : Float64[])
end

# iterative_smoother_solve(A, smoother, x0, b; maxiter = 1000, tol = 1e-6, alpha = 1.0) -> x, iter, res, err
# (src/solvers.jl:189-213) for a device smoother, the reference's full 4-tuple: A is the host matrix the reference
# passes (only the host fallback of the direct solve of :194 reads it), the sweeps run on S.A, and
# err[i] = ||x_i - A \\ b||_2 (:202) is formed on the device (aggmg_smoother_solve_dev).  exact = false: err empty.
function iterative_smoother_solve(A::sp.SparseMatrixCSC{Float64,Int64}, S::DeviceSmoother, x0::AbstractVector,
        b::AbstractVector; maxiter::Integer = 1000, tol::AbstractFloat = 1e-6, alpha::AbstractFloat = 1.0,
        exact::Bool = true, check_every::Integer = 1)
    N = length(b)
    dx0 = DeviceVector(S.ctx, x0); db = DeviceVector(S.ctx, b)
    dx = DeviceVector(S.ctx, N)
    nchk = cld(max(maxiter, 1), check_every)
    res = zeros(nchk); err = zeros(nchk); nit = Ref{Cint}(0); nck = Ref{Cint}(0)
    if exact
        ue = solve(direct_solver(S, S.A), A, db)
        GC.@preserve dx0 db dx ue res err check(S.ctx.h, ccall((:aggmg_smoother_solve_dev, LIB), Cint,
            (Handle, Handle, Handle, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Float64, Cint, Ptr{Cvoid},
             Ptr{Float64}, Ref{Cint}, Ref{Cint}, Ptr{Cvoid}, Ptr{Float64}),
            S.ctx.h, S.A.h, S.h, dx0.p, db.p, maxiter, Float64(tol), Float64(alpha), check_every, dx.p, res, nit, nck,
            ue.p, err))
    else
        GC.@preserve dx0 db dx res check(S.ctx.h, ccall((:aggmg_smoother_solve_dev, LIB), Cint,
            (Handle, Handle, Handle, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Float64, Cint, Ptr{Cvoid},
             Ptr{Float64}, Ref{Cint}, Ref{Cint}, Ptr{Cvoid}, Ptr{Float64}),
            S.ctx.h, S.A.h, S.h, dx0.p, db.p, maxiter, Float64(tol), Float64(alpha), check_every, dx.p, res, nit, nck,
            C_NULL, C_NULL))
    end
    return download(dx), Int(nit[]), res[1:nck[]], (exact ? err[1:nck[]] : Float64[])
end

end # module
