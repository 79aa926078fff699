"""Environment registration.

The reference imports `register` but never calls it — the call is commented out
(gym_soccer/__init__.py:3-12, id 'SoccerSimultaneous-v0', kwargs width=5,height=4,slip_prob=0.2,
max_episode_steps=100).  Here the registration exists and works with `gym`, with `gymnasium`, or with
neither installed (this image and the GPU box have neither): ids are kept in a local registry and
`make()` resolves them.  'SoccerLittman94-v0' is the id BASELINE.json uses for the same env with the
constructor's default slip_prob=0.0.
"""
ENV_SPECS = {
    "SoccerSimultaneous-v0": dict(entry_point="gym_soccer_littman94_amd.envs:SoccerSimultaneousEnv",
                                  kwargs={"width": 5, "height": 4, "slip_prob": 0.2,
                                          "player_a_policy": None, "player_b_policy": None}),
    "SoccerLittman94-v0": dict(entry_point="gym_soccer_littman94_amd.envs:SoccerSimultaneousEnv",
                               kwargs={"width": 5, "height": 4, "slip_prob": 0.0,
                                       "player_a_policy": None, "player_b_policy": None}),
    "VectorSoccerLittman94-v0": dict(entry_point="gym_soccer_littman94_amd.envs:VectorSoccerEnv",
                                     kwargs={"width": 5, "height": 4, "slip_prob": 0.0}),
}

_registered_with = []


def register_all():
    """Register with whichever of gymnasium / gym is importable; always fills the local registry."""
    for modname in ("gymnasium", "gym"):
        try:
            mod = __import__(modname + ".envs.registration", fromlist=["register"])
        except Exception:
            continue
        for env_id, spec in ENV_SPECS.items():
            if env_id.startswith("Vector"):
                continue
            try:
                mod.register(id=env_id, entry_point=spec["entry_point"], kwargs=dict(spec["kwargs"]),
                             max_episode_steps=100, reward_threshold=1.0, nondeterministic=True)
            except Exception:
                pass        # already registered
        _registered_with.append(modname)
    return list(_registered_with)


def make(env_id, **kwargs):
    """Local equivalent of gym.make for the ids above (no gym needed)."""
    if env_id not in ENV_SPECS:
        raise KeyError("unknown environment id %r; known: %s" % (env_id, ", ".join(sorted(ENV_SPECS))))
    spec = ENV_SPECS[env_id]
    modname, clsname = spec["entry_point"].split(":")
    cls = getattr(__import__(modname, fromlist=[clsname]), clsname)
    kw = dict(spec["kwargs"]); kw.update(kwargs)
    return cls(**kw)
