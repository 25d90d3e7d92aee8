"""Algorithm plugin table: `cfg.algo.name` ("Agent" + name for baselines) is looked up here by class name."""
from pql_amd.utils.common import ClassIndex

alg_name_to_path = ClassIndex(__file__)
