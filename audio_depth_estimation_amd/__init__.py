"""Import alias: the package lives in ``audio-depth-estimation_amd/`` (hyphenated, as the
project layout prescribes), which Python cannot import by name.  This stub re-points the
package search path there so that ``import audio_depth_estimation_amd.models...`` works."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'audio-depth-estimation_amd')
__path__ = [_real]
with open(_os.path.join(_real, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_real, '__init__.py'), 'exec'))
del _os, _f, _real
