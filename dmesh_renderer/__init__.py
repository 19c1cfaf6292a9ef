"""Drop-in alias: `import dmesh_renderer` (the reference package name) resolves to the MI355X build."""
from dmesh_renderer_amd import *  # noqa: F401,F403
from dmesh_renderer_amd import _C, __all__  # noqa: F401
