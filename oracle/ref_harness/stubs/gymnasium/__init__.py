"""Minimal stand-in for the `gymnasium` package (not installed in this image).

TEST INFRASTRUCTURE ONLY: lets `/root/reference/citylearn` import so that the
reference itself can be run as the parity oracle and golden-vector generator
(SURVEY.md App. C).  Nothing under `citylearn_amd/` imports this.
"""
from . import spaces  # noqa: F401


class Env:
    metadata = {}
    render_mode = None

    @property
    def unwrapped(self):
        return self

    def close(self):
        pass


class Wrapper(Env):
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith('_'):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped


class ObservationWrapper(Wrapper):
    pass


class ActionWrapper(Wrapper):
    pass


class RewardWrapper(Wrapper):
    pass
