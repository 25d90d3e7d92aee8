"""`pql` -- the reference's package name, answered by pql_amd.

north_star: "keeps the pql.algo / scripts/train_pql.py entry points".  Code written against the reference does
`import pql`, `pql.LIB_PATH / 'cfg'` (pql/__init__.py:3, scripts/train_pql.py:27) and
`from pql.algo.pql_v_learner import PQLVLearner`, `from pql.replay.simple_replay import ReplayBuffer`,
`from pql.utils.common import ...` (scripts/train_pql.py:8-24).  This package holds no code of its own: a
meta-path finder resolves every `pql.<x>` to the SAME module object as `pql_amd.<x>` (one copy of every class, so
`isinstance` and the class-name plugin tables agree whichever name a caller used).
"""
import importlib
import importlib.abc
import importlib.util
import sys

import pql_amd
from pql_amd import LIB_PATH, __version__  # noqa: F401

_REAL = "pql_amd"
# reference module names whose file is spelled differently here
_RENAMED = {"pql.algo.crossQ": "pql_amd.algo.crossq", "pql.utils.isaacgym_util": "pql_amd.envs.synthetic"}


class _Alias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith("pql."):
            return None
        real = _RENAMED.get(fullname, _REAL + fullname[3:])
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except ModuleNotFoundError:
            return None
        spec = importlib.util.spec_from_loader(fullname, self)
        spec._pql_real = real
        return spec

    def create_module(self, spec):
        return importlib.import_module(spec._pql_real)

    def exec_module(self, module):   # already executed under its real name
        pass


if not any(isinstance(f, _Alias) for f in sys.meta_path):
    sys.meta_path.insert(0, _Alias())

__path__ = []   # a package with no files of its own: every submodule comes from the finder above


def __getattr__(name):
    """`pql.algo`, `pql.models` ... without an explicit submodule import."""
    try:
        return importlib.import_module(f"pql.{name}")
    except ModuleNotFoundError as e:
        raise AttributeError(name) from e
