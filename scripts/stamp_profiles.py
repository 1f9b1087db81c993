"""Write the identity of the library the profiles of a round were taken with (tf_version, source digest, sha256 of the .so) next to them:
`python scripts/stamp_profiles.py profiles/r06_stamp.json`.  bench.py compares it with the library it loaded (`roofline.rocprof_stale`)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import _hip
json.dump(_hip.identity(), open(sys.argv[1], "w"), indent=1)
print(open(sys.argv[1]).read())
