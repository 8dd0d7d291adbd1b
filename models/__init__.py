"""Drop-in alias of the reference's top-level `models` package: with this repository on sys.path,
`import models.ULIP_models as models` (main_cls.py:25) and `from models.pointbert.point_encoder
import PointTransformer` resolve to the MI355X-native implementation in ppt_amd.models."""
