# Tests of the ROCm binding, written against the same independent check the reference's own test
# files use (sparse(A) * x with the COO triples of src/sparse.jl) -- NOT EXECUTED in this
# repository (no Julia toolchain in the build image); the Python suite tests/test_reference_suite_mirror.py
# runs the same statements through the same C entry points.
#
#   julia --project -e 'include("julia/test/runtests.jl")'      (needs BlockSparseMatrices, LinearMaps,
#                                                                  BSM_ROCM_LIB=/path/to/libbsmrocm.so)
using Test, LinearAlgebra, SparseArrays, Random
using BlockSparseMatrices
include(joinpath(@__DIR__, "..", "BlockSparseMatricesROCm.jl"))
using .BlockSparseMatricesROCm: ROCmScheduler

relmax(a, b) = maximum(abs.(a .- b)) / maximum(abs.(b))

@testset "VBCRS on the GPU ($T)" for T in (Float64, Float32, ComplexF64)
    Random.seed!(1)
    n = 6_000
    cuts = sort(unique(vcat(1, rand(2:n, 250), n + 1)))          # consecutive segments
    segs = [cuts[k]:(cuts[k + 1] - 1) for k in 1:(length(cuts) - 1)]
    pairs = unique([(rand(1:length(segs)), rand(1:length(segs))) for _ in 1:400])
    blocks = [randn(T, length(segs[i]), length(segs[j])) for (i, j) in pairs]
    r0 = [first(segs[i]) for (i, _) in pairs]
    c0 = [first(segs[j]) for (_, j) in pairs]
    cpu = VariableBlockCompressedRowStorage(blocks, r0, c0, (n, n))
    gpu = VariableBlockCompressedRowStorage(blocks, r0, c0, (n, n); scheduler=ROCmScheduler())
    @test gpu.rowptr == cpu.rowptr && gpu.colindices == cpu.colindices && gpu.rowindices == cpu.rowindices
    S = sparse(cpu)
    tol = T === Float32 ? 1f-5 : 1e-12
    for _ in 1:5
        x = randn(T, n); y = randn(T, n)
        @test relmax(gpu * x, S * x) < tol
        @test relmax(gpu' * x, S' * x) < tol
        @test relmax(transpose(gpu) * x, transpose(S) * x) < tol
        α, β = T(0.5), T(-2)
        @test relmax(mul!(copy(y), gpu, x, α, β), α * (S * x) + β * y) < tol
        @test relmax(mul!(fill(T(NaN), n), gpu, x, true, false), S * x) < tol   # strong zero
        X = randn(T, n, 5)
        @test relmax(gpu * X, S * X) < tol                                       # bsm_mul_multi
    end
    @test nnz(gpu) == nnz(cpu)
end

@testset "BlockSparseMatrix / SymmetricBlockMatrix on the GPU" begin
    Random.seed!(2)
    n = 3_000
    lists() = sort(randperm(n)[1:rand(3:28)])
    blocks = Matrix{ComplexF64}[]
    ri = Vector{Vector{Int}}(); ci = Vector{Vector{Int}}()
    for _ in 1:200
        r, c = lists(), lists()
        push!(blocks, randn(ComplexF64, length(r), length(c))); push!(ri, r); push!(ci, c)
    end
    A = BlockSparseMatrix(blocks, ri, ci, (n, n); scheduler=ROCmScheduler())
    S = sparse(BlockSparseMatrix(blocks, ri, ci, (n, n)))
    for _ in 1:5
        x = randn(ComplexF64, n); y = randn(ComplexF64, n)
        @test A * x ≈ S * x
        @test A' * x ≈ S' * x
        @test transpose(A) * x ≈ transpose(S) * x
        @test mul!(copy(y), A, x, im, 2im) ≈ im * (S * x) + 2im * y
    end
    # symmetric: disjoint diagonal index sets, off-diagonal blocks between different sets
    perm = randperm(n); sets = [sort(perm[(20k + 1):(20k + 20)]) for k in 0:(n ÷ 20 - 1)]
    D = [(d = randn(ComplexF64, 20, 20); (d + transpose(d)) / 2) for _ in sets]
    offp = unique([(i, rand(1:(i - 1))) for i in rand(2:length(sets), 300)])
    O = [randn(ComplexF64, 20, 20) for _ in offp]
    sym(s) = SymmetricBlockMatrix(D, sets, O, [sets[i] for (i, _) in offp], [sets[j] for (_, j) in offp], (n, n); scheduler=s)
    G = sym(ROCmScheduler()); Sg = sparse(sym(BlockSparseMatrices.SerialScheduler()))
    @test issymmetric(Sg)
    for _ in 1:5
        x = randn(ComplexF64, n); y = randn(ComplexF64, n)
        @test G * x ≈ Sg * x
        @test G' * x ≈ Sg' * x
        @test mul!(copy(y), G, x, im, 2im) ≈ im * (Sg * x) + 2im * y
    end
    @test nnz(G) == nnz(Sg)
end

@testset "fallbacks, page-locked vectors, converters, several (virtual) devices" begin
    using .BlockSparseMatricesROCm: ROCmVBCRS, pin!
    Random.seed!(3)
    n = 2_000
    segs = [(20k + 1):(20k + 20) for k in 0:(n ÷ 20 - 1)]
    pairs = unique([(rand(1:length(segs)), rand(1:length(segs))) for _ in 1:300])
    blocks = [randn(Float64, 20, 20) for _ in pairs]
    r0 = [first(segs[i]) for (i, _) in pairs]; c0 = [first(segs[j]) for (_, j) in pairs]
    S = sparse(VariableBlockCompressedRowStorage(blocks, r0, c0, (n, n)))
    A = VariableBlockCompressedRowStorage(blocks, r0, c0, (n, n); scheduler=ROCmScheduler())
    A2 = VariableBlockCompressedRowStorage(blocks, r0, c0, (n, n); scheduler=ROCmScheduler(devices=[0, 0]))
    X = randn(n, 3); x = pin!(randn(n)); y = pin!(zeros(n))
    @test mul!(y, A, x) ≈ S * x                                  # DMA straight from / to the pinned vectors
    @test A2 * x ≈ S * x                                         # two parts, halo-free row partition
    @test A * view(X, :, 2) ≈ S * X[:, 2]                        # SubArray -> contiguous temporary
    xc = randn(ComplexF64, n); yc = randn(ComplexF64, n)
    @test mul!(copy(yc), A, xc, im, 2im) ≈ im * (S * xc) + 2im * yc   # complex x, α, β on a real matrix
    # converters: same bookkeeping as the reference's, transposes not materialised
    sets = [collect(s) for s in segs]
    D = [(d = randn(20, 20); (d + d') / 2) for _ in sets]
    offp = unique([(i, rand(1:(i - 1))) for i in rand(2:length(sets), 200)])
    O = [randn(20, 20) for _ in offp]
    sym = SymmetricBlockMatrix(D, sets, O, [sets[i] for (i, _) in offp], [sets[j] for (_, j) in offp], (n, n))
    ref = VariableBlockCompressedRowStorage(sym)
    V = ROCmVBCRS(sym)
    @test V.rowptr == ref.rowptr && V.colindices == ref.colindices && V.rowindices == ref.rowindices
    @test V * x ≈ sparse(sym) * x
    @test V' * x ≈ sparse(sym)' * x
end
