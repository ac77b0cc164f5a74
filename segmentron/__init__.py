"""Drop-in name.  With this repository ahead of the reference checkout on sys.path,
`import segmentron` serves the hot path (`segmentron.config`, `segmentron.models[.model_zoo,
.backbones, ...]`, `segmentron.modules`, `segmentron.utils.registry`) from segmentron_amd — the
HIP kernels — and every other `segmentron.*` module (solver, data pipeline, utils, options) from
the reference checkout's own files, so tools/train.py, tools/eval.py and tools/demo.py run
unchanged.  All logic lives in segmentron_amd/dropin.py."""
import sys

from segmentron_amd import dropin as _dropin

_dropin.install(sys.modules[__name__])
