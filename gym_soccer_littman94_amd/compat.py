"""Import-name compatibility with the reference package.

`install_as_gym_soccer()` registers this package's classes under the reference's module names
(`gym_soccer`, `gym_soccer.envs`, `gym_soccer.envs.soccer_simultaneous_env`, `gym_soccer.utils.planners`,
`gym_soccer.utils.policies`), so code written against the reference — including its test files — runs
unchanged:

    import gym_soccer_littman94_amd as gsa; gsa.install_as_gym_soccer()
    from gym_soccer.envs import SoccerSimultaneousEnv            # the MI355X drop-in
    from gym_soccer.utils.planners import value_iteration        # one HIP kernel
"""
import sys
import types


def install_as_gym_soccer(force=False):
    if "gym_soccer" in sys.modules and not force and not getattr(sys.modules["gym_soccer"], "_amd_alias", False):
        raise RuntimeError("a gym_soccer package is already imported; pass force=True to shadow it")
    from . import planners, policies
    from .envs import soccer_env
    from .envs.soccer_env import SoccerSimultaneousEnv

    def module(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m._amd_alias = True
        sys.modules[name] = m
        return m

    root = module("gym_soccer", __path__=[])
    envs = module("gym_soccer.envs", __path__=[], SoccerSimultaneousEnv=SoccerSimultaneousEnv)
    envs.soccer_simultaneous_env = module("gym_soccer.envs.soccer_simultaneous_env",
                                          **{k: v for k, v in vars(soccer_env).items() if not k.startswith("__")})
    utils = module("gym_soccer.utils", __path__=[])
    utils.planners = module("gym_soccer.utils.planners", **{k: v for k, v in vars(planners).items() if not k.startswith("__")})
    utils.policies = module("gym_soccer.utils.policies", **{k: v for k, v in vars(policies).items() if not k.startswith("__")})
    root.envs, root.utils = envs, utils
    return root
