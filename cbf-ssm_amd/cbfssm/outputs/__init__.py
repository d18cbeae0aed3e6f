from .outputs import Outputs
from .output_summary import OutputSummary

# OutputsRoboMove / OutputsVoliro (reference outputs/__init__.py:2-3) add dataset-specific plots only; they are
# reporting code outside the hot path and are aliased to the generic Outputs here.
OutputsRoboMove = Outputs
