# SimulatedAnnealingABCHIP.jl -- thin Julia binding of libsabc_hip.so (include/sabc_hip.h).
#
# Keeps the reference's public surface for the hot path -- `sabc`, `update_population!`,
# `SABCresult`, `SABCstate`, `RandomWalk`, `DifferentialEvolution`, `StretchMove`
# (src/SimulatedAnnealingABC.jl:19,28-60,251-259,451-460; src/proposals.jl:6) -- and does
# nothing but marshal arguments into `ccall`s.  All numerics live in the library.
#
# NOT EXECUTED IN THE BUILD CONTAINER: Julia is not installed there (SURVEY.md section 8c).  The
# Python mirror in ../api.py makes exactly the same calls and is what the test-suite drives.
module SimulatedAnnealingABCHIP

using Distributions: Distribution, Normal, Uniform, Exponential, LogNormal, Product, UnivariateDistribution
import Base: show

export sabc, update_population!, RandomWalk, DifferentialEvolution, StretchMove,
       DeviceDistance, GaussianIID, Gaussian2D, GandK, LotkaVolterra

const libsabc = get(ENV, "SABC_HIP_LIB", joinpath(@__DIR__, "..", "libsabc_hip.so"))

const MAX_PARA, MAX_STATS, MAX_MODEL_PARAMS = 8, 8, 32

# ---- C structs (must match include/sabc_hip.h field for field) ----
struct CConfig
    abi_version::Int32
    device::Int32
    n_particles::Int64
    n_para::Int32
    n_stats::Int32
    model_id::Int32
    n_model_params::Int32
    model_params::NTuple{MAX_MODEL_PARAMS,Float64}
    prior_kind::NTuple{MAX_PARA,Int32}
    prior_a::NTuple{MAX_PARA,Float64}
    prior_b::NTuple{MAX_PARA,Float64}
    algorithm::Int32
    rank::Int32
    world::Int32
    reserved::Int32
    v::Float64
    delta::Float64
    seed::UInt64
end

struct CUpdateArgs
    n_simulation::Int64
    v::Float64
    delta::Float64
    resample::Float64
    checkpoint_history::Int64
    proposal_kind::Int32
    reserved::Int32
    proposal_p0::Float64
    proposal_p1::Float64
end

# ---- proposals: same constructors and errors as src/proposals.jl ----
abstract type Proposal end

mutable struct RandomWalk{T} <: Proposal
    β::Float64
    Σ::T
end
function RandomWalk(; β=0.8, n_para)
    (0 < β <= 1) || error("Mixing parameter `β` must be between zero and one.")   # proposals.jl:30
    n_para == 1 ? RandomWalk(β, -1.0) : RandomWalk(β, -ones(n_para, n_para))
end

struct DifferentialEvolution <: Proposal
    γ0::Float64
    σ_gamma::Float64
    function DifferentialEvolution(; γ0=nothing, n_para=nothing, σ_gamma=1e-5)
        if !isnothing(γ0) && isnothing(n_para)
            new(γ0, σ_gamma)
        elseif !isnothing(n_para) && isnothing(γ0)
            new(2.38 / sqrt(2 * n_para), σ_gamma)                                  # proposals.jl:93
        else
            throw(ArgumentError("Provide either `γ0` or `n_para`, not both."))   # proposals.jl:96
        end
    end
end

struct StretchMove <: Proposal
    a::Float64
end
StretchMove(; a=2) = StretchMove(a)

descriptor(p::RandomWalk) = (Int32(0), p.β, 0.0)
descriptor(p::DifferentialEvolution) = (Int32(1), p.γ0, p.σ_gamma)
descriptor(p::StretchMove) = (Int32(2), p.a, 0.0)

# ---- device-coded simulators: `f_dist` as data (<: Function so that sabc(f_dist::Function, ...) dispatches) ----
abstract type DeviceDistance <: Function end
struct GaussianIID <: DeviceDistance
    n_obs::Int; sd::Float64; obs_mean::Float64; obs_m2::Union{Nothing,Float64}
end
GaussianIID(; n_obs=100, sd=1.0, obs_mean=0.0, obs_m2=nothing) = GaussianIID(n_obs, sd, obs_mean, obs_m2)
struct Gaussian2D <: DeviceDistance
    n_obs::Int; r::Float64; obs_mean::NTuple{2,Float64}; obs_varsum::Float64; obs_cov::Float64
end
struct GandK <: DeviceDistance
    n_draws::Int; c::Float64; ranks::NTuple{4,Int}; obs::NTuple{4,Float64}
end
struct LotkaVolterra <: DeviceDistance
    n_steps::Int; dt::Float64; σ::Float64; x0::Float64; y0::Float64; obs::NTuple{4,Float64}
end
model_id(::GaussianIID) = Int32(1); model_id(::Gaussian2D) = Int32(2)
model_id(::GandK) = Int32(3); model_id(::LotkaVolterra) = Int32(4)
n_stats(m::GaussianIID) = isnothing(m.obs_m2) ? 1 : 2
n_stats(::Gaussian2D) = 3; n_stats(::GandK) = 4; n_stats(::LotkaVolterra) = 4
params(m::GaussianIID) = Float64[m.n_obs, m.sd, m.obs_mean, something(m.obs_m2, 0.0)]
params(m::Gaussian2D) = Float64[m.n_obs, m.r, m.obs_mean..., m.obs_varsum, m.obs_cov]
params(m::GandK) = Float64[m.n_draws, m.c, m.ranks..., m.obs...]
params(m::LotkaVolterra) = Float64[m.n_steps, m.dt, m.σ, m.x0, m.y0, m.obs...]

# ---- any other f_dist: stays a Julia function, called back by the library for the proposals that
#      passed the prior gate (include/sabc_hip.h: sabc_simulate_fn); everything else runs on the GPU
struct HostDistance{F} <: DeviceDistance
    f::F
    n_stats::Int
    n_para::Int
    args::Tuple
    kwargs::NamedTuple
end
model_id(::HostDistance) = Int32(0)
n_stats(m::HostDistance) = m.n_stats
params(::HostDistance) = Float64[]

function host_callback(m::HostDistance)
    function cb(ctx::Ptr{Cvoid}, theta::Ptr{Float64}, ids::Ptr{Int64}, n::Int64, iter::UInt64, rho::Ptr{Float64})::Cint
        try
            Θ = unsafe_wrap(Array, theta, (Int(n), m.n_para))        # column-major n x d
            R = unsafe_wrap(Array, rho, (Int(n), m.n_stats))
            Threads.@threads for i in 1:n                             # like SimulatedAnnealingABC.jl:308
                θ = m.n_para == 1 ? Θ[i, 1] : Θ[i, :]
                R[i, :] .= collect(Float64, m.f(θ, m.args...; m.kwargs...))
            end
            return Cint(0)
        catch
            return Cint(-1)
        end
    end
    @cfunction($cb, Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}, Int64, UInt64, Ptr{Float64}))
end

# ---- priors as data: Normal, Uniform and products of those ----
prior_descriptor(d::Normal) = (Int32(0), d.μ, d.σ)
prior_descriptor(d::Uniform) = (Int32(1), d.a, d.b)
prior_descriptor(d::Exponential) = (Int32(2), d.θ, 0.0)
prior_descriptor(d::LogNormal) = (Int32(3), d.μ, d.σ)
prior_descriptors(d::UnivariateDistribution) = [prior_descriptor(d)]
prior_descriptors(d::Product) = [prior_descriptor(c) for c in d.v]

# ---- result types: same field names as SimulatedAnnealingABC.jl:28-60 ----
mutable struct SABCstate
    ϵ::Vector{Float64}
    algorithm::Symbol
    ϵ_history::Vector{Vector{Float64}}
    ρ_history::Vector{Vector{Float64}}
    u_history::Vector{Vector{Float64}}
    cdfs_dist_prior
    n_simulation::Int
    n_accept::Int
    n_resampling::Int
    n_population_updates::Int
end

struct SABCresult{T,S}
    population::Vector{T}
    u::Array{S}
    ρ::Array{S}
    state::SABCstate
    handle::Base.RefValue{Ptr{Cvoid}}     # owns the device-resident population
end

function check(h::Ptr{Cvoid}, rc::Integer)
    rc == 0 && return
    msg = unsafe_string(h == C_NULL ? ccall((:sabc_last_global_error, libsabc), Cstring, ()) :
                                      ccall((:sabc_last_error, libsabc), Cstring, (Ptr{Cvoid},), h))
    error(msg)                           # ErrorException, like the reference's error(...) sites
end

padtuple(v, n, T) = ntuple(i -> i <= length(v) ? T(v[i]) : zero(T), n)

function create_handle(f_dist::DeviceDistance, prior; n_particles, algorithm, v, δ, seed, device=0)
    pd = prior_descriptors(prior)
    p = params(f_dist)
    cfg = Ref(CConfig(1, device, n_particles, length(pd), n_stats(f_dist), model_id(f_dist), length(p),
                      padtuple(p, MAX_MODEL_PARAMS, Float64),
                      padtuple(first.(pd), MAX_PARA, Int32),
                      padtuple(getindex.(pd, 2), MAX_PARA, Float64), padtuple(last.(pd), MAX_PARA, Float64),
                      algorithm == :multi_eps ? 1 : 0, 0, 1, 0, v, δ, seed))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(C_NULL, ccall((:sabc_create, libsabc), Cint, (Ref{CConfig}, Ref{Ptr{Cvoid}}), cfg, h))
    finalizer(r -> (r[] != C_NULL && ccall((:sabc_destroy, libsabc), Cvoid, (Ptr{Cvoid},), r[]); r[] = C_NULL), h)
    if f_dist isa HostDistance
        cb = host_callback(f_dist)
        HOST_CALLBACKS[h[]] = (cb, f_dist)             # keep the closure (and the wrapped model) alive with the handle
        check(h[], ccall((:sabc_set_host_simulator, libsabc), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h[], cb, C_NULL))
    end
    h
end
const HOST_CALLBACKS = Dict{Ptr{Cvoid},Any}()

# the reference's own signature (SimulatedAnnealingABC.jl:451): any Function; the number of statistics is
# probed with one call on a prior draw, as at :163-165
function sabc(f_dist::Function, prior::Distribution, args...; kwargs...)
    f_dist isa DeviceDistance && return invoke(sabc, Tuple{DeviceDistance,Distribution}, f_dist, prior; kwargs...)
    own = (:n_particles, :n_simulation, :algorithm, :proposal, :resample, :v, :δ, :checkpoint_history,
           :show_progressbar, :show_checkpoint, :seed, :device)
    mine = (; (k => v for (k, v) in kwargs if k in own)...)
    theirs = (; (k => v for (k, v) in kwargs if !(k in own))...)
    ρ = f_dist(rand(prior), args...; theirs...)
    hd = HostDistance(f_dist, length(ρ), length(prior), args, theirs)
    invoke(sabc, Tuple{DeviceDistance,Distribution}, hd, prior; mine...)
end

# copies device state into the Julia arrays in place (`.=` at SimulatedAnnealingABC.jl:395-397)
function refresh!(res::SABCresult, d::Int, s::Int)
    h = res.handle[]
    n = length(res.population)
    θ = Matrix{Float64}(undef, n, d)               # column-major n x d == the library's [d][n]
    GC.@preserve θ check(h, ccall((:sabc_get_population, libsabc), Cint,
                                  (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h, θ, res.u, res.ρ))
    if eltype(res.population) <: Real
        res.population .= vec(θ)
    else
        for i in 1:n
            res.population[i] = θ[i, :]
        end
    end
    st = res.state
    eps = Vector{Float64}(undef, MAX_STATS); len = Ref{Int32}(0)
    check(h, ccall((:sabc_get_epsilon, libsabc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ref{Int32}), h, eps, len))
    st.ϵ = eps[1:len[]]
    c = Vector{Int64}(undef, 4)
    check(h, ccall((:sabc_get_counters, libsabc), Cint, (Ptr{Cvoid}, Ptr{Int64}), h, c))
    st.n_simulation, st.n_accept, st.n_resampling, st.n_population_updates = c
    m = ccall((:sabc_history_len, libsabc), Int64, (Ptr{Cvoid},), h)
    le = length(st.ϵ)
    eh = Matrix{Float64}(undef, le, m); uh = Matrix{Float64}(undef, s, m); rh = Matrix{Float64}(undef, s, m)
    check(h, ccall((:sabc_get_history, libsabc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h, eh, uh, rh))
    st.ϵ_history = [eh[:, k] for k in 1:m]; st.u_history = [uh[:, k] for k in 1:m]; st.ρ_history = [rh[:, k] for k in 1:m]
    res
end

"""
    update_population!(population_state, f_dist, prior; n_simulation, v=1.0, δ=0.1, proposal, resample, checkpoint_history=1)

Same keywords as SimulatedAnnealingABC.jl:251-259.  Mutates and returns `population_state`.
"""
function update_population!(res::SABCresult, f_dist::DeviceDistance, prior::Distribution;
                            n_simulation, v=1.0, δ=0.1,
                            proposal::Proposal=DifferentialEvolution(n_para=length(prior)),
                            resample=2 * length(res.population), checkpoint_history=1,
                            show_progressbar::Bool=false, show_checkpoint=Inf)
    v <= 0 && error("Annealing speed `v` must be positive.")                       # :261
    δ <= 0 && error("Resamping intensity `δ` must be positive.")                   # :262
    h = res.handle[]
    d, s = length(prior), size(res.u, 2)
    θ = eltype(res.population) <: Real ? reshape(copy(res.population), :, 1) : permutedims(reduce(hcat, res.population))
    GC.@preserve θ check(h, ccall((:sabc_set_population, libsabc), Cint,
                                  (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h, θ, res.u, res.ρ))
    kind, p0, p1 = descriptor(proposal)
    args = Ref(CUpdateArgs(n_simulation, v, δ, resample, checkpoint_history, kind, 0, p0, p1))
    check(h, ccall((:sabc_update, libsabc), Cint, (Ptr{Cvoid}, Ref{CUpdateArgs}), h, args))
    if proposal isa RandomWalk
        Σ = Matrix{Float64}(undef, d, d)
        check(h, ccall((:sabc_get_proposal_sigma, libsabc), Cint, (Ptr{Cvoid}, Ptr{Float64}), h, Σ))
        d == 1 ? (proposal.Σ = Σ[1, 1]) : (proposal.Σ .= Σ)
    end
    refresh!(res, d, s)
end

# update_population!(res, f_dist, prior, args...; kwargs...) with the plain function the result was created
# with (SimulatedAnnealingABC.jl:251): look the wrapped model up by handle; args/kwargs go to f_dist (:315)
function update_population!(res::SABCresult, f_dist::Function, prior::Distribution, args...; kwargs...)
    f_dist isa DeviceDistance && return invoke(update_population!, Tuple{SABCresult,DeviceDistance,Distribution}, res, f_dist, prior; kwargs...)
    haskey(HOST_CALLBACKS, res.handle[]) || error("this SABCresult was not created with a host `f_dist`")
    hd = HOST_CALLBACKS[res.handle[]][2]
    hd.f === f_dist || error("`f_dist` differs from the one this SABCresult was initialised with")
    own = (:n_simulation, :v, :δ, :proposal, :resample, :checkpoint_history, :show_progressbar, :show_checkpoint)
    mine = (; (k => v for (k, v) in kwargs if k in own)...)
    invoke(update_population!, Tuple{SABCresult,DeviceDistance,Distribution}, res, hd, prior; mine...)
end

"""
    sabc(f_dist::DeviceDistance, prior; n_particles=100, n_simulation=10_000, algorithm=:single_eps, ...)

Same keywords as SimulatedAnnealingABC.jl:451-460 (+ `seed`, `device`).
"""
function sabc(f_dist::DeviceDistance, prior::Distribution;
              n_particles=100, n_simulation=10_000, algorithm=:single_eps,
              proposal::Proposal=DifferentialEvolution(n_para=length(prior)),
              resample=2 * n_particles, v=1.0, δ=0.1, checkpoint_history=1,
              show_progressbar::Bool=false, show_checkpoint=Inf, seed=rand(UInt64) >> 1, device=0)
    (algorithm == :multi_eps || algorithm == :single_eps) ||
        error("Argument `algorithm` must be :multi_eps or :single_eps, not `$algorithm`!")   # :462-464
    n_simulation < n_particles &&
        error("`n_simulation = $n_simulation` is too small for $n_particles particles.")     # :155-156
    h = create_handle(f_dist, prior; n_particles, algorithm, v, δ, seed, device)
    check(h[], ccall((:sabc_initialize, libsabc), Cint, (Ptr{Cvoid}, Int64), h[], n_simulation))
    d, s = length(prior), n_stats(f_dist)
    T = d == 1 ? Float64 : Vector{Float64}
    pop = d == 1 ? zeros(n_particles) : [zeros(d) for _ in 1:n_particles]
    cdf = ρ -> begin
        out = Vector{Float64}(undef, s)
        r = collect(Float64, ρ)
        check(h[], ccall((:sabc_cdf_apply, libsabc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}), h[], r, 1, out))
        out
    end
    st = SABCstate(Float64[], algorithm, [], [], [], cdf, 0, 0, 0, 0)
    res = SABCresult{T,Float64}(pop, zeros(n_particles, s), zeros(n_particles, s), st, h)
    refresh!(res, d, s)
    n_sim_remaining = n_simulation - st.n_simulation                                         # :478
    n_sim_remaining < n_particles && @warn "`n_simulation` too small to update all particles!"
    update_population!(res, f_dist, prior; n_simulation=n_sim_remaining, resample, proposal, v, δ, checkpoint_history)
end

function show(io::IO, s::SABCresult)     # SimulatedAnnealingABC.jl:65-82
    n = length(s.population)
    println(io, "Approximate posterior sample with $n particles:")
    println(io, "  - algorithm: :$(s.state.algorithm)")
    println(io, "  - simulations used: $(s.state.n_simulation)")
    println(io, "  - number of population updates: $(s.state.n_population_updates)")
    println(io, "  - ϵ: $(round.(s.state.ϵ, sigdigits=4))")
    println(io, "  - number of population resamplings: $(s.state.n_resampling)")
    println(io, "  - acceptance rate: $(round(s.state.n_accept / (s.state.n_simulation - n), sigdigits=4))")
end

end # module
