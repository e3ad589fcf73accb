"""neuralplane_amd — MI355X-native fused F-16 env.step for NeuralPlane-style fixed-wing RL.

Hot path only: per-aircraft FDM step (43-MLP aero model, 6-DoF EoM, atmosphere,
Euler/RK4) + observation + reward + termination as ONE hand-written HIP kernel per `env.step`,
behind the reference's `ControlEnv.reset()/step()` surface (neuralplane_amd.envs).
"""
__version__ = '0.1.0'
