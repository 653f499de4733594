"""Type aliases the reference imports for annotations only."""
from typing import Any

_MAP_LOCATION_TYPE = Any
_PATH = Any
