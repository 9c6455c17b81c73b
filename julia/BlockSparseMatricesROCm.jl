# BlockSparseMatricesROCm.jl -- reference-side binding of libbsmrocm.so (include/bsm_rocm.h).
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: no Julia toolchain exists in the build image.  It is the
# stub a maintainer of BlockSparseMatrices.jl would add (e.g. as a package extension): it opts a
# matrix into the MI355X path through the EXISTING `scheduler=` keyword, so no reference signature
# changes.  The Python mirror (blocksparsematrices.jl_amd/matrices.py) exercises exactly the same
# C entry points and is what the parity tests run.
module BlockSparseMatricesROCm

using LinearAlgebra, LinearMaps
using BlockSparseMatrices
import BlockSparseMatrices: AbstractBlockMatrix, BlockSparseMatrix, SymmetricBlockMatrix,
    VariableBlockCompressedRowStorage

const libbsm = get(ENV, "BSM_ROCM_LIB", "libbsmrocm.so")

"Opt-in scheduler: `BlockSparseMatrix(...; scheduler=ROCmScheduler())`."
struct ROCmScheduler
    device::Int32            # HIP ordinal, -1 = current device
    accumulate::Int32        # 0 auto, 1 atomics, 2 coloured launches, 3 gather, 4 direct (2-4: bitwise reproducible)
    transpose_image::Int32   # 1: keep a second, transposed ordering for A' / transpose(A)
end
ROCmScheduler(; device=-1, accumulate=0, transpose_image=0) = ROCmScheduler(device, accumulate, transpose_image)
BlockSparseMatrices.isserial(::ROCmScheduler) = true   # no host colouring needed for the GPU path

mutable struct BsmOptions           # mirrors bsm_options (72 bytes)
    struct_size::Int32; device::Int32; scheduler::Int32; accumulate::Int32
    validate::Int32; transpose_image::Int32; own_lo::Int64; own_hi::Int64
    reserved::NTuple{4,Int64}
end

function _check(rc)
    rc == 0 || error("libbsmrocm: " * unsafe_string(ccall((:bsm_last_error, libbsm), Cstring, ())))
end

const _DTYPE = Dict(Float32 => 0, Float64 => 1, ComplexF32 => 2, ComplexF64 => 3)

mutable struct Handle
    ptr::Ptr{Cvoid}
    function Handle(p)
        h = new(p)
        finalizer(x -> ccall((:bsm_destroy, libbsm), Cint, (Ptr{Cvoid},), x.ptr), h)
    end
end

const _handles = WeakKeyDict{Any,Handle}()

function _options(s::ROCmScheduler)
    o = Ref(BsmOptions(0, 0, 0, 0, 0, 0, 0, 0, (0, 0, 0, 0)))
    ccall((:bsm_options_default, libbsm), Cvoid, (Ref{BsmOptions},), o)
    o[].device = s.device
    o[].accumulate = s.accumulate
    o[].transpose_image = s.transpose_image
    return o
end

# replaces the analysis done by the constructor src/vbcrs.jl:78-122 + the loop :266-288
function handle(A::VariableBlockCompressedRowStorage{T}) where {T}
    get!(_handles, A) do
        nb = length(A.blocks)
        m = Int64[size(b, 1) for b in A.blocks]; n = Int64[size(b, 2) for b in A.blocks]
        rowstart = Int64[A.rowindices[searchsortedlast(A.rowptr, i)] for i in 1:nb]
        colstart = Int64.(A.colindices)
        ptrs = Ptr{Cvoid}[pointer(b) for b in A.blocks]
        out = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve A _check(ccall((:bsm_vbcrs_create, libbsm), Cint,
            (Cint, Int64, Int64, Int64, Ptr{Ptr{Cvoid}}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64},
             Ptr{Int64}, Ptr{Int64}, Ref{BsmOptions}, Ref{Ptr{Cvoid}}),
            _DTYPE[T], size(A, 1), size(A, 2), nb, ptrs, m, n, m, rowstart, colstart,
            _options(A.scheduler), out))
        Handle(out[])
    end
end

# replaces src/blockmatrix.jl:62-109 (+ :225-247)
function handle(A::BlockSparseMatrix{T}) where {T}
    get!(_handles, A) do
        nb = length(A.blocks)
        m = Int64[size(b, 1) for b in A.blocks]; n = Int64[size(b, 2) for b in A.blocks]
        ri = [Int64.(r) for r in A.rowindices]; ci = [Int64.(c) for c in A.colindices]
        out = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve A ri ci _check(ccall((:bsm_blocksparse_create, libbsm), Cint,
            (Cint, Int64, Int64, Int64, Ptr{Ptr{Cvoid}}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64},
             Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ref{BsmOptions}, Ref{Ptr{Cvoid}}),
            _DTYPE[T], size(A, 1), size(A, 2), nb, Ptr{Cvoid}[pointer(b) for b in A.blocks], m, n, m,
            pointer.(ri), pointer.(ci), _options(A.scheduler), out))
        Handle(out[])
    end
end

# replaces src/symmetricblockmatrix.jl:73-126 (+ :386-435)
function handle(A::SymmetricBlockMatrix{T}) where {T}
    get!(_handles, A) do
        ds = Int64[size(b, 1) for b in A.diagonals]
        m = Int64[size(b, 1) for b in A.offdiagonals]; n = Int64[size(b, 2) for b in A.offdiagonals]
        di = [Int64.(d) for d in A.diagonalindices]
        ri = [Int64.(r) for r in A.rowindices]; ci = [Int64.(c) for c in A.colindices]
        out = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve A di ri ci _check(ccall((:bsm_symmetric_create, libbsm), Cint,
            (Cint, Int64, Int64, Int64, Ptr{Ptr{Cvoid}}, Ptr{Int64}, Ptr{Int64}, Ptr{Ptr{Int64}},
             Int64, Ptr{Ptr{Cvoid}}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Ptr{Int64}},
             Ptr{Ptr{Int64}}, Ref{BsmOptions}, Ref{Ptr{Cvoid}}),
            _DTYPE[T], size(A, 1), size(A, 2), length(ds), Ptr{Cvoid}[pointer(b) for b in A.diagonals],
            ds, ds, pointer.(di), length(m), Ptr{Cvoid}[pointer(b) for b in A.offdiagonals], m, n, m,
            pointer.(ri), pointer.(ci), _options(A.scheduler), out))
        Handle(out[])
    end
end

const ROCmMat = Union{BlockSparseMatrix{<:Any,<:Any,<:Any,ROCmScheduler},
    SymmetricBlockMatrix{<:Any,<:Any,<:Any,<:Any,ROCmScheduler},
    VariableBlockCompressedRowStorage{<:Any,<:Any,<:Any,ROCmScheduler}}

_op(::ROCmMat) = 0
_op(::LinearMaps.TransposeMap) = 1
_op(::LinearMaps.AdjointMap) = 2
_base(A::ROCmMat) = A
_base(A) = A.lmap

# the drop-in: same signature as src/blockmatrix.jl:225, src/symmetricblockmatrix.jl:386,
# src/vbcrs.jl:266,343.  beta === false is Julia's strong zero (src/abstractblockmatrix.jl:27-34).
function LinearMaps._unsafe_mul!(y::Vector{T}, A::Union{Z,LinearMaps.AdjointMap{<:Any,Z},
        LinearMaps.TransposeMap{<:Any,Z}}, x::Vector{T}, α::Number, β::Number) where {T,Z<:ROCmMat}
    h = handle(_base(A))
    a = Ref(T(α)); b = Ref(T(β === false ? 0 : β))
    GC.@preserve x y _check(ccall((:bsm_mul, libbsm), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ref{T}, Ref{T}, Cint, Cint, Ptr{Cvoid}),
        h.ptr, _op(A), x, y, a, b, β === false, 0 #= BSM_MEM_HOST =#, C_NULL))
    return y
end
# device-resident vectors (AMDGPU.jl ROCVector): identical call with memspace = 1 and the
# task-local HIP stream instead of C_NULL.

# `A * X` / mul!(Y, A, X, α, β) with matrices: LinearMaps would loop the columns through the
# method above (one sweep of A per column); bsm_mul_multi streams A once per 8 columns.
function LinearMaps._unsafe_mul!(Y::Matrix{T}, A::Union{Z,LinearMaps.AdjointMap{<:Any,Z},
        LinearMaps.TransposeMap{<:Any,Z}}, X::Matrix{T}, α::Number, β::Number) where {T,Z<:ROCmMat}
    h = handle(_base(A))
    a = Ref(T(α)); b = Ref(T(β === false ? 0 : β))
    GC.@preserve X Y _check(ccall((:bsm_mul_multi, libbsm), Cint,
        (Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ref{T}, Ref{T}, Cint, Cint, Ptr{Cvoid}),
        h.ptr, _op(A), size(X, 2), X, stride(X, 2), Y, stride(Y, 2), a, b, β === false, 0, C_NULL))
    return Y
end

# VariableBlockCompressedRowStorage(sbm) without materialising transpose(offdiagonals)
# (reference src/vbcrs.jl:189-264): bsm_vbcrs_create_from_symmetric with first(...) of every list.

end # module
