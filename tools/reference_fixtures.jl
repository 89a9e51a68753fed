#!/usr/bin/env julia
# reference_fixtures.jl -- the recipe that pins the oracle to the REAL reference, for whoever has Julia.
#
#   julia --project=<env with SimulatedAnnealingABC v0.4.0> tools/reference_fixtures.jl [outdir = tests/golden]
#
# Julia is not installed in the build container, so this script has never run there and the files it writes
# (tests/golden/reference_*.json) are absent from the repository: tests/test_reference_fixtures.py reports
# "parity unpinned" until they exist and compares the oracle (CPU) and the device operators (GPU) with them once they do.
#
# Everything dumped here is DETERMINISTIC -- no rand(), no threads: the reference's own functions on fixed inputs
#   build_cdf(::Vector), build_cdf(::Matrix)           src/cdf_estimators.jl:23-44,58-73   (Interpolations.jl)
#   update_epsilon_single_eps / _multi_eps             src/SimulatedAnnealingABC.jl:92-117 (Roots.jl)
#   resample_population (its ESS: the weights)         src/SimulatedAnnealingABC.jl:124-137
#   update_proposal!(::RandomWalk, population)         src/proposals.jl:46-48,58-60        (StatsBase.cov)
# Inputs are written next to the outputs, so the tests need nothing but the JSON files.
using SimulatedAnnealingABC
const SABC = SimulatedAnnealingABC

outdir = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "tests", "golden")
mkpath(outdir)

# ---- a tiny JSON writer (no extra package): numbers with round-trip precision, nested arrays, flat dicts ----
js(x::AbstractFloat) = isfinite(x) ? repr(Float64(x)) : (isnan(x) ? "\"NaN\"" : (x > 0 ? "\"Inf\"" : "\"-Inf\""))
js(x::Integer) = string(x)
js(x::AbstractString) = "\"" * x * "\""
js(x::Symbol) = js(string(x))
js(x::Union{AbstractVector,Tuple}) = "[" * join((js(e) for e in x), ", ") * "]"
js(x::AbstractMatrix) = js([collect(r) for r in eachrow(x)])                 # list of rows
js(d::AbstractDict) = "{" * join(("\"$(k)\": " * js(v) for (k, v) in d), ", ") * "}"
write_json(name, d) = open(io -> println(io, js(d)), joinpath(outdir, name), "w")

# Weyl sequence: deterministic, irregular, reproducible anywhere as frac(i * phi) in exact binary64 arithmetic
weyl(i) = (x = i * 0.6180339887498949; x - floor(x))

# ---- 1. build_cdf(::Vector): the three inputs of test/runtests.jl:11,18,24 (the random one replaced by a Weyl vector) ----
cdf_cases = Any[]
for (label, x) in (("weyl100_times4", [4 * weyl(i) for i in 1:100]),
                   ("repeats", Float64[1, 2, 2, 3, 3, 3]),
                   ("zeros", Float64[1, 0, 2, 0, 3]))
    cdf = SABC.build_cdf(x)
    top = 1.5 * maximum(x)
    q = vcat([-1.0, 0.0, 1e-300, top, top * 1.0000001, 1e300],          # flat ends, the first knot, the last knot
             sort(unique(x)),                                            # every knot value (duplicates included once)
             [top * weyl(k + 1000) for k in 1:(50 - 6)])                 # between knots
    push!(cdf_cases, Dict("label" => label, "x" => x, "q" => q, "cdf" => [cdf(qi) for qi in q]))
end
# build_cdf(::Matrix) + closure: 3 statistics, 40 particles
M = hcat([3 * weyl(i) for i in 1:40], [weyl(i + 77)^2 for i in 1:40], [i % 5 == 0 ? 0.0 : 10 * weyl(i + 500) for i in 1:40])
F = SABC.build_cdf(M)
rows = [M[i, :] for i in 1:5:40]
write_json("reference_cdf.json", Dict("vector_cases" => cdf_cases,
           "matrix" => Dict("rho" => M, "query_rows" => rows, "u" => [F(r) for r in rows])))

# ---- 2. epsilon schedules ----
single = Any[]
for ubar in (1e-17, 1e-6, 0.01, 0.1, 0.3, 0.5, 0.9), v in (0.1, 1.0, 10.0)
    push!(single, Dict("ubar" => ubar, "v" => v, "eps" => SABC.update_epsilon_single_eps(ubar, v)))
end
multi = Any[]
for ubar in ([0.4], [0.05], [0.3, 0.2], [0.5, 0.01], [0.45, 0.3, 0.1], [0.02, 0.03, 0.04], [0.49, 0.4, 0.3, 0.2], [0.1, 0.1, 0.1, 0.1]),
    v in (0.3, 1.0, 5.0)
    u = vcat(reshape(0.5 .* ubar, 1, :), reshape(1.5 .* ubar, 1, :))       # two particles whose column means are ubar
    push!(multi, Dict("ubar" => ubar, "u" => u, "v" => v, "eps" => SABC.update_epsilon_multi_eps(u, v)))
end
write_json("reference_epsilon.json", Dict("single_eps" => single, "multi_eps" => multi))

# ---- 3. resample_population: the effective sample size pins the weights w = exp(-sum_j u_ij delta / ubar_j) ----
res = Any[]
for (n, s, δ) in ((50, 1, 0.1), (50, 3, 0.1), (200, 2, 0.7))
    u = [weyl(i + 31 * j)^(j) for i in 1:n, j in 1:s]
    pop = collect(1.0:n)
    _, _, ess = SABC.resample_population(pop, u, δ)
    push!(res, Dict("u" => u, "delta" => δ, "ess" => ess))
end
write_json("reference_resample.json", Dict("cases" => res))

# ---- 4. update_proposal!(::RandomWalk): Sigma = beta (cov + 1e-8 I), 1-D: beta var ----
prop = Any[]
let n = 64
    pop1 = [3 * weyl(i) - 1 for i in 1:n]
    rw1 = SABC.RandomWalk(β=0.8, n_para=1)
    SABC.update_proposal!(rw1, pop1)
    push!(prop, Dict("d" => 1, "beta" => 0.8, "population" => reshape(pop1, :, 1), "sigma" => [[rw1.Σ]]))
    for (d, β) in ((2, 0.8), (4, 0.35))
        pop = [[(k + 1) * weyl(i + 13 * k) + 0.3 * weyl(i) for k in 1:d] for i in 1:n]
        rw = SABC.RandomWalk(β=β, n_para=d)
        SABC.update_proposal!(rw, pop)
        push!(prop, Dict("d" => d, "beta" => β, "population" => permutedims(reduce(hcat, pop)), "sigma" => rw.Σ))
    end
end
write_json("reference_proposal.json", Dict("cases" => prop))

println("wrote reference_cdf.json, reference_epsilon.json, reference_resample.json, reference_proposal.json to ", outdir)
