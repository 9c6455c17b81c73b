# BlockSparseMatricesROCm.jl -- reference-side binding of libbsmrocm.so (include/bsm_rocm.h).
#
# EXPERIMENTAL / NOT EXECUTED IN THIS REPOSITORY: no Julia toolchain exists in the build image, so
# this file has never run.  It is the binding a maintainer of BlockSparseMatrices.jl would add (e.g.
# as a package extension): a matrix opts into the MI355X path through the EXISTING `scheduler=`
# keyword, so no reference signature changes.  The Python mirror
# (blocksparsematrices.jl_amd/matrices.py) exercises exactly the same C entry points with the same
# argument conventions and is what the parity tests run.
module BlockSparseMatricesROCm

using LinearAlgebra, LinearMaps
using BlockSparseMatrices
import BlockSparseMatrices: AbstractBlockMatrix, BlockSparseMatrix, SymmetricBlockMatrix,
    VariableBlockCompressedRowStorage

const libbsm = get(ENV, "BSM_ROCM_LIB", "libbsmrocm.so")

# ---- the opt-in scheduler ---------------------------------------------------------------------------
"""
    ROCmScheduler(; device=-1, devices=Int32[], accumulate=0, transpose_image=2)

`BlockSparseMatrix(...; scheduler=ROCmScheduler())` etc.  `devices = [0, 1, ...]` spreads ONE matrix
over several GPUs of the node (bsm_ctx_t: block rows partitioned by stored bytes, halo exchange over
xGMI) -- the counterpart of the reference's `@tasks` fan-out (src/vbcrs.jl:275-276).
"""
struct ROCmScheduler
    device::Int32            # HIP ordinal, -1 = current device
    devices::Vector{Int32}   # non-empty: multi-GPU handle over these ordinals
    accumulate::Int32        # 0 auto, 1 atomics, 2 coloured launches, 3 gather, 4 direct (2-4: bitwise reproducible)
    transpose_image::Int32   # 1: keep a second, transposed ordering for A' / transpose(A); 2: when it is cheap
end
ROCmScheduler(; device=-1, devices=Int32[], accumulate=0, transpose_image=2) =
    ROCmScheduler(Int32(device), Int32.(collect(devices)), Int32(accumulate), Int32(transpose_image))
BlockSparseMatrices.isserial(::ROCmScheduler) = true   # no host colouring needed for the GPU path

mutable struct BsmOptions           # mirrors bsm_options (72 bytes)
    struct_size::Int32; device::Int32; scheduler::Int32; accumulate::Int32
    validate::Int32; transpose_image::Int32; own_lo::Int64; own_hi::Int64
    ctx::Ptr{Cvoid}; blocks_memspace::Int64; coloring::Int64
    reserved::NTuple{1,Int64}
end

function _check(rc)
    rc == 0 || error("libbsmrocm: " * unsafe_string(ccall((:bsm_last_error, libbsm), Cstring, ())))
    return nothing
end

const _DTYPE = Dict(Float32 => 0, Float64 => 1, ComplexF32 => 2, ComplexF64 => 3)
const ROCmEltype = Union{Float32,Float64,ComplexF32,ComplexF64}

mutable struct Handle
    ptr::Ptr{Cvoid}
    function Handle(p)
        h = new(p)
        finalizer(x -> ccall((:bsm_destroy, libbsm), Cint, (Ptr{Cvoid},), x.ptr), h)
    end
end

# contexts of devices live as long as the process (handles keep a pointer into them)
const _ctxs = Dict{Vector{Int32},Ptr{Cvoid}}()
const _lock = ReentrantLock()
function _ctx(devices::Vector{Int32})
    lock(_lock) do
        get!(_ctxs, devices) do
            out = Ref{Ptr{Cvoid}}(C_NULL)
            _check(ccall((:bsm_ctx_create, libbsm), Cint, (Ptr{Int32}, Int32, Ref{Ptr{Cvoid}}),
                         devices, length(devices), out))
            out[]
        end
    end
end

function _options(s::ROCmScheduler)
    o = Ref(BsmOptions(0, 0, 0, 0, 0, 0, 0, 0, C_NULL, 0, 0, (0,)))
    ccall((:bsm_options_default, libbsm), Cvoid, (Ref{BsmOptions},), o)
    o[].device = s.device
    o[].accumulate = s.accumulate
    o[].transpose_image = isempty(s.devices) ? s.transpose_image : Int32(0)
    isempty(s.devices) || (o[].ctx = _ctx(s.devices))
    return o
end

# ---- handle cache -------------------------------------------------------------------------------------
# The reference's three matrix types are IMMUTABLE structs: they cannot be keys of a WeakKeyDict
# (Julia refuses to attach a finalizer to them).  Every constructor allocates at least one fresh
# mutable Vector per instance (rowptr / colors / diagonalcolors): that vector is the key, so the
# handle dies with the matrix and two matrices never share one.
const _handles = WeakKeyDict{Any,Handle}()
_key(A::VariableBlockCompressedRowStorage) = A.rowptr
_key(A::BlockSparseMatrix) = A.colors
_key(A::SymmetricBlockMatrix) = A.diagonalcolors
handle(A) = lock(() -> get!(() -> _create(A), _handles, _key(A)), _lock)

_ld(b) = Int64(max(stride(b, 2), size(b, 1), 1))   # leading dimension; >= 1 also for 0-row blocks

# A block the C ABI can take: element type T, column-major with unit row stride.  The reference admits any
# AbstractMatrix as a block and counts it as prod(size) (_nnz, src/abstractblockmatrix.jl:65-71): sparse blocks,
# adjoints, views with a row stride are densified ONCE here, at handle creation.
_cblock(::Type{T}, b::StridedMatrix{T}) where {T} = stride(b, 1) == 1 ? b : Matrix{T}(b)
_cblock(::Type{T}, b::AbstractMatrix) where {T} = Matrix{T}(b)
_cblocks(::Type{T}, bs) where {T} = [_cblock(T, b) for b in bs]

# replaces the analysis done by the constructor src/vbcrs.jl:78-122 + the loop :266-288
function _create(A::VariableBlockCompressedRowStorage{T}) where {T}
    nb = length(A.blocks)
    m = Int64[size(b, 1) for b in A.blocks]; n = Int64[size(b, 2) for b in A.blocks]
    rowstart = Int64[A.rowindices[searchsortedlast(A.rowptr, i)] for i in 1:nb]
    colstart = Int64.(A.colindices)
    bl = _cblocks(T, A.blocks)
    ptrs = Ptr{Cvoid}[pointer(b) for b in bl]
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve A bl _check(ccall((:bsm_vbcrs_create, libbsm), Cint,
        (Cint, Int64, Int64, Int64, Ptr{Ptr{Cvoid}}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64},
         Ptr{Int64}, Ptr{Int64}, Ref{BsmOptions}, Ref{Ptr{Cvoid}}),
        _DTYPE[T], size(A, 1), size(A, 2), nb, ptrs, m, n, _ld.(bl), rowstart, colstart,
        _options(A.scheduler), out))
    return Handle(out[])
end

# replaces src/blockmatrix.jl:62-109 (+ :225-247)
function _create(A::BlockSparseMatrix{T}) where {T}
    nb = length(A.blocks)
    m = Int64[size(b, 1) for b in A.blocks]; n = Int64[size(b, 2) for b in A.blocks]
    ri = [Vector{Int64}(r) for r in A.rowindices]; ci = [Vector{Int64}(c) for c in A.colindices]
    bl = _cblocks(T, A.blocks)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve A bl ri ci _check(ccall((:bsm_blocksparse_create, libbsm), Cint,
        (Cint, Int64, Int64, Int64, Ptr{Ptr{Cvoid}}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64},
         Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ref{BsmOptions}, Ref{Ptr{Cvoid}}),
        _DTYPE[T], size(A, 1), size(A, 2), nb, Ptr{Cvoid}[pointer(b) for b in bl], m, n,
        _ld.(bl), pointer.(ri), pointer.(ci), _options(A.scheduler), out))
    return Handle(out[])
end

# replaces src/symmetricblockmatrix.jl:73-126 (+ :386-435)
function _create(A::SymmetricBlockMatrix{T}) where {T}
    ds = Int64[size(b, 1) for b in A.diagonals]
    m = Int64[size(b, 1) for b in A.offdiagonals]; n = Int64[size(b, 2) for b in A.offdiagonals]
    di = [Vector{Int64}(d) for d in A.diagonalindices]
    ri = [Vector{Int64}(r) for r in A.rowindices]; ci = [Vector{Int64}(c) for c in A.colindices]
    dg = _cblocks(T, A.diagonals); og = _cblocks(T, A.offdiagonals)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve A dg og di ri ci _check(ccall((:bsm_symmetric_create, libbsm), Cint,
        (Cint, Int64, Int64, Int64, Ptr{Ptr{Cvoid}}, Ptr{Int64}, Ptr{Int64}, Ptr{Ptr{Int64}},
         Int64, Ptr{Ptr{Cvoid}}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Ptr{Int64}},
         Ptr{Ptr{Int64}}, Ref{BsmOptions}, Ref{Ptr{Cvoid}}),
        _DTYPE[T], size(A, 1), size(A, 2), length(ds), Ptr{Cvoid}[pointer(b) for b in dg],
        ds, _ld.(dg), pointer.(di), length(m), Ptr{Cvoid}[pointer(b) for b in og],
        m, n, _ld.(og), pointer.(ri), pointer.(ci), _options(A.scheduler), out))
    return Handle(out[])
end

const ROCmMat = Union{BlockSparseMatrix{<:Any,<:Any,<:Any,ROCmScheduler},
    SymmetricBlockMatrix{<:Any,<:Any,<:Any,<:Any,ROCmScheduler},
    VariableBlockCompressedRowStorage{<:Any,<:Any,<:Any,ROCmScheduler}}
const ROCmOp{Z} = Union{Z,LinearMaps.AdjointMap{<:Any,Z},LinearMaps.TransposeMap{<:Any,Z}}

_op(::ROCmMat) = 0
_op(::LinearMaps.TransposeMap) = 1
_op(::LinearMaps.AdjointMap) = 2
_base(A::ROCmMat) = A
_base(A) = A.lmap

# ---- the drop-in ----------------------------------------------------------------------------------------
# Same signature as src/blockmatrix.jl:225, src/symmetricblockmatrix.jl:386, src/vbcrs.jl:266,343.
# beta === false is Julia's strong zero (src/abstractblockmatrix.jl:27-34).
function _mul!(y, A, x, α::T, β::T, strong::Bool, memspace::Integer, stream::Ptr{Cvoid}, ::Type{T}) where {T}
    h = handle(_base(A))
    a = Ref(α); b = Ref(β)
    GC.@preserve x y _check(ccall((:bsm_mul, libbsm), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ref{T}, Ref{T}, Cint, Cint, Ptr{Cvoid}),
        h.ptr, _op(A), pointer(x), pointer(y), a, b, strong, memspace, stream))
    return y
end

_fits(::Type{T}, s::Number) where {T} = s isa Bool || s isa T || (T <: Complex) || !(s isa Complex)

# fast path: plain host vectors of the matrix' element type (BSM_MEM_HOST; page-lock long-lived
# vectors with `pin!` to make both PCIe copies true DMA)
function LinearMaps._unsafe_mul!(y::Vector{T}, A::ROCmOp{Z}, x::Vector{T}, α::Number, β::Number) where
        {T<:ROCmEltype,Z<:ROCmMat}
    if eltype(_base(A)) === T && _fits(T, α) && _fits(T, β)
        return _mul!(y, A, x, T(α), T(β === false ? 0 : β), β === false, 0, C_NULL, T)
    end
    return _fallback_mul!(y, A, x, α, β)
end

# The 3-argument form.  The reference defines its own for the VBCRS wrappers (src/vbcrs.jl:331-341:
# `fill!(y, zero(T))`, then the 5-argument method with β = true) -- for a ROCmScheduler that would zero y on
# the host, upload it and add: forward Julia's strong zero instead (y never travels to the device).
function LinearMaps._unsafe_mul!(y::AbstractVector, A::ROCmOp{Z}, x::AbstractVector) where {Z<:ROCmMat}
    return LinearMaps._unsafe_mul!(y, A, x, true, false)
end

# everything else LinearMaps may hand over -- SubArray columns of `A * X`, strided views, other
# element types, complex α / β on a real matrix: by linearity through contiguous Vector{T} temporaries.
# NEVER the reference's own loop: its `@tasks ... @set scheduler = ...` cannot run a ROCmScheduler.
function LinearMaps._unsafe_mul!(y::AbstractVector, A::ROCmOp{Z}, x::AbstractVector, α::Number, β::Number) where
        {Z<:ROCmMat}
    return _fallback_mul!(y, A, x, α, β)
end

function _fallback_mul!(y, A, x, α, β)
    T = eltype(_base(A))
    # (one result temporary per product; `convert` copies x only when it is not already a Vector{T})
    gpu(v) = _mul!(Vector{T}(undef, size(A, 1)), A, convert(Vector{T}, v), one(T), zero(T), true, 0, C_NULL, T)
    t = (T <: Real && eltype(x) <: Complex) ? complex.(gpu(real.(x)), gpu(imag.(x))) : gpu(x)
    if β === false
        y .= α .* t
    else
        y .= α .* t .+ β .* y
    end
    return y
end

# `A * X` / mul!(Y, A, X, α, β) with matrices: LinearMaps would loop the columns through the vector
# method (one sweep of A per column); bsm_mul_multi streams A once per 8 columns.
function LinearMaps._unsafe_mul!(Y::Matrix{T}, A::ROCmOp{Z}, X::Matrix{T}, α::Number, β::Number) where
        {T<:ROCmEltype,Z<:ROCmMat}
    if !(eltype(_base(A)) === T && _fits(T, α) && _fits(T, β))
        for k in axes(X, 2)
            _fallback_mul!(view(Y, :, k), A, view(X, :, k), α, β)
        end
        return Y
    end
    h = handle(_base(A))
    a = Ref(T(α)); b = Ref(T(β === false ? 0 : β))
    GC.@preserve X Y _check(ccall((:bsm_mul_multi, libbsm), Cint,
        (Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ref{T}, Ref{T}, Cint, Cint, Ptr{Cvoid}),
        h.ptr, _op(A), size(X, 2), X, max(stride(X, 2), 1), Y, max(stride(Y, 2), 1), a, b, β === false, 0, C_NULL))
    return Y
end

"Page-lock a long-lived host vector used as x / y (bsm_host_register); undone by its finalizer."
function pin!(v::Vector)
    _check(ccall((:bsm_host_register, libbsm), Cint, (Ptr{Cvoid}, Int64), v, sizeof(v)))
    finalizer(w -> ccall((:bsm_host_unregister, libbsm), Cint, (Ptr{Cvoid},), w), v)
    return v
end

# mirrors bsm_part_info_t (include/bsm_rocm.h)
struct BsmPartInfo
    device::Int32; pad::Int32
    own_lo::Int64; own_hi::Int64; touched_lo::Int64; touched_hi::Int64
    device_bytes::Int64; nblocks::Int64; col_lo::Int64; col_hi::Int64
    reserved::NTuple{2,Int64}
end

"""
    part_ranges(A) -> Vector{(device, rows, cols)}

The parts of a matrix spread over several GPUs (`ROCmScheduler(devices=[...])`): part p lives on `device`, owns the y
entries `rows` and holds the x entries `cols` of a partitioned product (bsm_part_info; 1-based inclusive ranges).
"""
function part_ranges(A)
    h = handle(_base(A))
    n = length(_base(A).scheduler.devices)   # one part per listed device
    out = NamedTuple{(:device, :rows, :cols),Tuple{Int32,UnitRange{Int64},UnitRange{Int64}}}[]
    for p in 0:n-1
        info = Ref(BsmPartInfo(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, (0, 0)))
        _check(ccall((:bsm_part_info, libbsm), Cint, (Ptr{Cvoid}, Int32, Ref{BsmPartInfo}), h.ptr, p, info))
        push!(out, (device=info[].device, rows=info[].own_lo:info[].own_hi, cols=info[].col_lo:info[].col_hi))
    end
    return out
end

# ---- device-resident vectors (AMDGPU.jl) -----------------------------------------------------------------
# Iterative solvers keep x / y in HBM: BSM_MEM_DEVICE, enqueued on the task-local HIP stream, no
# synchronisation (the product is 10 us for a C2-sized operator against 74-118 us through host vectors).
# Loaded only where AMDGPU.jl is installed.
if Base.find_package("AMDGPU") !== nothing
    @eval begin
        import AMDGPU
        function LinearMaps._unsafe_mul!(y::AMDGPU.ROCVector{T}, A::ROCmOp{Z}, x::AMDGPU.ROCVector{T},
                α::Number, β::Number) where {T<:ROCmEltype,Z<:ROCmMat}
            (eltype(_base(A)) === T && _fits(T, α) && _fits(T, β)) ||
                throw(ArgumentError("device vectors must have the matrix' element type; α, β convertible to it"))
            st = Base.unsafe_convert(Ptr{Cvoid}, AMDGPU.stream().stream)
            return _mul!(y, A, x, T(α), T(β === false ? 0 : β), β === false, 1, st, T)
        end

        """
            mul_parts!(yparts, A, xparts, α=true, β=false)

        `mul!` for a matrix spread over several GPUs (`ROCmScheduler(devices=[...])`) with x and y PARTITIONED
        like its block rows: `xparts[p]` / `yparts[p]` are `ROCVector`s on device `devices[p]` holding the
        entries of part p's column / row range (`part_ranges(A)`).  Only the halo a part reads beyond its own
        slice and the y segments it produced for rows of another device travel (bsm_mul_parts, xGMI).
        """
        function mul_parts!(yparts::Vector{<:AMDGPU.ROCVector{T}}, A::ROCmOp{Z}, xparts::Vector{<:AMDGPU.ROCVector{T}},
                α::Number=true, β::Number=false) where {T<:ROCmEltype,Z<:ROCmMat}
            h = handle(_base(A))
            # the C side reads one pointer per part and trusts the part lengths: check both here
            pr = part_ranges(A)
            (length(xparts) == length(pr) && length(yparts) == length(pr)) ||
                throw(DimensionMismatch("mul_parts!: \$(length(pr)) parts, got \$(length(xparts)) x parts and \$(length(yparts)) y parts"))
            for (p, r) in enumerate(pr)
                # include/bsm_rocm.h at bsm_mul_parts: op N reads the column range and delivers the row range, op T / C the reverse
                xr, yr = _op(A) == 0 ? (r.cols, r.rows) : (r.rows, r.cols)
                (length(xparts[p]) == length(xr) && length(yparts[p]) == length(yr)) ||
                    throw(DimensionMismatch("mul_parts!: part \$p holds x[\$(xr)] and y[\$(yr)]"))
            end
            xp = Ptr{Cvoid}[Base.unsafe_convert(Ptr{Cvoid}, pointer(v)) for v in xparts]
            yp = Ptr{Cvoid}[Base.unsafe_convert(Ptr{Cvoid}, pointer(v)) for v in yparts]
            a = Ref(T(α)); b = Ref(T(β === false ? 0 : β))
            GC.@preserve xparts yparts _check(ccall((:bsm_mul_parts, libbsm), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ref{T}, Ref{T}, Cint, Ptr{Ptr{Cvoid}}),
                h.ptr, _op(A), xp, yp, a, b, β === false, C_NULL))   # NULL streams: every device's default stream
            return yparts
        end
    end
end

# ---- converters (reference src/vbcrs.jl:150-264) -----------------------------------------------------------
"""
    ROCmVBCRS(A::BlockSparseMatrix | A::SymmetricBlockMatrix)

The reference's `VariableBlockCompressedRowStorage(bsm)` / `(sbm)` converters on the GPU.  For a
SymmetricBlockMatrix the reference materialises `transpose.(offdiagonals)` (twice the off-diagonal
storage, src/vbcrs.jl:222-262); here the bookkeeping (`rowptr`, `colindices`, `rowindices` over the
`ndiag + 2 noff` virtual blocks) is identical -- fetched from the library, bit-exact -- but every
off-diagonal block is stored and streamed once (bsm_vbcrs_create_from_symmetric).
"""
struct ROCmVBCRS{T} <: LinearMaps.LinearMap{T}
    handle::Handle
    size::Tuple{Int,Int}
    rowptr::Vector{Int64}
    colindices::Vector{Int64}
    rowindices::Vector{Int64}
    key::Vector{Int}          # fresh per instance (see _key)
end
Base.size(A::ROCmVBCRS) = A.size
_key(A::ROCmVBCRS) = A.key
handle(A::ROCmVBCRS) = A.handle
_base(A::ROCmVBCRS) = A
_op(::ROCmVBCRS) = 0

function _bookkeeping(h::Handle, which::Integer)
    len = Ref{Int64}(0)
    _check(ccall((:bsm_get_bookkeeping, libbsm), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}, Ref{Int64}), h.ptr, which, C_NULL, len))
    out = Vector{Int64}(undef, len[])
    _check(ccall((:bsm_get_bookkeeping, libbsm), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}, Ref{Int64}), h.ptr, which, out, len))
    return out
end

function ROCmVBCRS(A::SymmetricBlockMatrix{T}; scheduler::ROCmScheduler=ROCmScheduler()) where {T<:ROCmEltype}
    ds = Int64[size(b, 1) for b in A.diagonals]
    m = Int64[size(b, 1) for b in A.offdiagonals]; n = Int64[size(b, 2) for b in A.offdiagonals]
    d0 = Int64[first(d) for d in A.diagonalindices]        # first(...): src/vbcrs.jl:231-239
    r0 = Int64[first(r) for r in A.rowindices]; c0 = Int64[first(c) for c in A.colindices]
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve A _check(ccall((:bsm_vbcrs_create_from_symmetric, libbsm), Cint,
        (Cint, Int64, Int64, Int64, Ptr{Ptr{Cvoid}}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64},
         Int64, Ptr{Ptr{Cvoid}}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64},
         Ref{BsmOptions}, Ref{Ptr{Cvoid}}),
        _DTYPE[T], size(A, 1), size(A, 2), length(ds), Ptr{Cvoid}[pointer(b) for b in A.diagonals],
        ds, _ld.(A.diagonals), d0, length(m), Ptr{Cvoid}[pointer(b) for b in A.offdiagonals],
        m, n, _ld.(A.offdiagonals), r0, c0, _options(scheduler), out))
    h = Handle(out[])
    return ROCmVBCRS{T}(h, (size(A, 1), size(A, 2)), _bookkeeping(h, 1), _bookkeeping(h, 2), _bookkeeping(h, 3), Int[])
end

function ROCmVBCRS(A::BlockSparseMatrix{T}; scheduler::ROCmScheduler=ROCmScheduler()) where {T<:ROCmEltype}
    nb = length(A.blocks)
    m = Int64[size(b, 1) for b in A.blocks]; n = Int64[size(b, 2) for b in A.blocks]
    ri = [Vector{Int64}(r) for r in A.rowindices]; ci = [Vector{Int64}(c) for c in A.colindices]
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve A ri ci _check(ccall((:bsm_vbcrs_create_from_blocksparse, libbsm), Cint,
        (Cint, Int64, Int64, Int64, Ptr{Ptr{Cvoid}}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64},
         Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ref{BsmOptions}, Ref{Ptr{Cvoid}}),
        _DTYPE[T], size(A, 1), size(A, 2), nb, Ptr{Cvoid}[pointer(b) for b in A.blocks], m, n,
        _ld.(A.blocks), pointer.(ri), pointer.(ci), _options(scheduler), out))
    h = Handle(out[])
    return ROCmVBCRS{T}(h, (size(A, 1), size(A, 2)), _bookkeeping(h, 1), _bookkeeping(h, 2), _bookkeeping(h, 3), Int[])
end

const ROCmViewOp = Union{ROCmVBCRS,LinearMaps.AdjointMap{<:Any,<:ROCmVBCRS},LinearMaps.TransposeMap{<:Any,<:ROCmVBCRS}}
function LinearMaps._unsafe_mul!(y::AbstractVector, A::ROCmViewOp, x::AbstractVector, α::Number, β::Number)
    T = eltype(_base(A))
    if y isa Vector{T} && x isa Vector{T} && _fits(T, α) && _fits(T, β)
        return _mul!(y, A, x, T(α), T(β === false ? 0 : β), β === false, 0, C_NULL, T)
    end
    return _fallback_mul!(y, A, x, α, β)
end
LinearMaps._unsafe_mul!(y::AbstractVector, A::ROCmViewOp, x::AbstractVector) =
    LinearMaps._unsafe_mul!(y, A, x, true, false)

end # module
