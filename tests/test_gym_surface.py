"""The Gymnasium surface of `CityLearnEnv` (reference citylearn.py:52): a `gymnasium.Env` subclass when gymnasium is importable."""
import importlib
import sys
import types


def test_env_is_a_gymnasium_env_when_gymnasium_is_importable(monkeypatch):
    """gymnasium is not part of this image, so a stand-in module is put in its place: `CityLearnEnv` must then derive from its `Env` (what
    `gymnasium.Wrapper.__init__` -- and with it every wrapper of the reference -- asserts) and leave `spec` to gymnasium."""
    import citylearn_amd.citylearn as mod
    import citylearn_amd.spaces as spaces_mod

    class Env:
        metadata = {'render_modes': []}
        spec = None

    gym = types.ModuleType('gymnasium')
    gym.Env = Env
    gym_spaces = types.ModuleType('gymnasium.spaces')
    gym_spaces.Box = spaces_mod.Box
    gym.spaces = gym_spaces
    monkeypatch.setitem(sys.modules, 'gymnasium', gym)
    monkeypatch.setitem(sys.modules, 'gymnasium.spaces', gym_spaces)
    try:
        reloaded = importlib.reload(mod)
        assert issubclass(reloaded.CityLearnEnv, Env)
        assert 'spec' not in reloaded.CityLearnEnv.__dict__ and reloaded.CityLearnEnv.spec is None
        assert {'reset', 'step', 'close', 'unwrapped', 'action_space', 'observation_space'} <= set(dir(reloaded.CityLearnEnv))
    finally:
        monkeypatch.undo()
        importlib.reload(mod)
    assert mod.CityLearnEnv.__mro__[1] is object                   # back to the plain class without gymnasium
    assert isinstance(mod.CityLearnEnv.__dict__['spec'], property)
