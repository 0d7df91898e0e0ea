"""TEST INFRASTRUCTURE ONLY -- the slice of gym 0.26 the reference touches
(rl_env/WRSN.py:3-4,21,31-32,299): `gym.Env` as a base class and `spaces.Box`
with `.low/.high/.shape/.dtype`.  gym is not installable in this image."""
from . import spaces  # noqa: F401


class Env:
    pass
