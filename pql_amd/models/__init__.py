"""Model plugin table: `cfg.algo.act_class` / `cfg.algo.cri_class` are looked up here by class name."""
from pql_amd.utils.common import ClassIndex

model_name_to_path = ClassIndex(__file__)
