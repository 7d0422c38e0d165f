"""Import shim: the package directory is `rs-face-detection-tflite_amd/` (hyphenated, as the project layout requires),
which Python cannot import by name.  `import rs_face_detection_tflite_amd` loads that directory as a package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rs-face-detection-tflite_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
