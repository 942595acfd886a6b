# Same import surface as pvn3d/lib/pointnet2_utils/__init__.py:8-10
from . import _ext
from . import pointnet2_utils
from . import pointnet2_modules
