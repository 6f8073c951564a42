"""`contrast.flow`: the online flow source of the pre-training step (RAFT inference), SURVEY.md §8f row 4."""
from .raft import RAFT, coords_grid  # noqa: F401
from ..util import upflow8  # noqa: F401
