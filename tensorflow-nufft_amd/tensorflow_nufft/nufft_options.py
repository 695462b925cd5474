"""`Options` for `nufft` -- mirrors the reference's user-visible options tree
(tensorflow_nufft/python/ops/nufft_options.py:25-273): `max_batch_size`,
`points_range`, `debugging.check_points_range`, `fftw.planning_rigor`, with
validation on assignment and a protobuf round trip that keeps the reference's
wire format (proto/nufft_options.proto:27-32). `fftw.*` is accepted and has no
effect on the GPU path (rocFFT plans are not tunable that way)."""
import enum
import typing

import pydantic

from tensorflow_nufft import _proto


class FftwPlanningRigor(enum.IntEnum):
  """Planning rigor of the reference's CPU FFT library; kept for API parity."""
  AUTO = 0
  ESTIMATE = 1
  MEASURE = 2
  PATIENT = 3
  EXHAUSTIVE = 4

  def to_proto(self):
    return int(self)

  @classmethod
  def from_proto(cls, pb):
    try:
      return cls(int(pb))
    except ValueError:
      raise ValueError(
          "Invalid value of `FftwPlanningRigor` in protocol buffer. Supported "
          "values include `AUTO`, `ESTIMATE`, `MEASURE`, `PATIENT` and "
          f"`EXHAUSTIVE`. Got {pb}.") from None


class PointsRange(enum.IntEnum):
  """Supported range of the non-uniform points: STRICT [-pi, pi], EXTENDED
  [-3 pi, 3 pi] (default), INFINITE (any). Folding rules: reference
  cc/kernels/nufft_plan.h:676-734."""
  STRICT = 0
  EXTENDED = 1
  INFINITE = 2

  def to_proto(self):
    return int(self)

  @classmethod
  def from_proto(cls, pb):
    try:
      return cls(int(pb))
    except ValueError:
      raise ValueError(
          "Invalid value of `PointsRange` in protocol buffer. Supported "
          f"values include `STRICT`, `EXTENDED` and `INFINITE`. Got {pb}.") from None


class _Model(pydantic.BaseModel):
  model_config = pydantic.ConfigDict(validate_assignment=True, extra='forbid')


class DebuggingOptions(_Model):
  """`check_points_range`: assert that the points lie within `points_range`."""
  check_points_range: bool = False

  def to_proto(self):
    return _proto.DebuggingOptionsProto(self.check_points_range)

  @classmethod
  def from_proto(cls, pb):
    return cls(check_points_range=pb.check_points_range)


class FftwOptions(_Model):
  """Options of the reference's CPU FFT backend (no effect here)."""
  planning_rigor: FftwPlanningRigor = FftwPlanningRigor.AUTO

  def to_proto(self):
    return _proto.FftwOptionsProto(self.planning_rigor.to_proto())

  @classmethod
  def from_proto(cls, pb):
    return cls(planning_rigor=FftwPlanningRigor.from_proto(pb.planning_rigor))


class Options(_Model):
  """Options for `nufft`.

  Attributes:
    debugging: `DebuggingOptions`.
    fftw: `FftwOptions` (accepted, ignored on the GPU).
    max_batch_size: maximum number of transforms processed per batch
      (`None` = automatic).
    points_range: `PointsRange`, default `EXTENDED`.
  """
  debugging: DebuggingOptions = pydantic.Field(default_factory=DebuggingOptions)
  fftw: FftwOptions = pydantic.Field(default_factory=FftwOptions)
  max_batch_size: typing.Optional[int] = None
  points_range: PointsRange = PointsRange.EXTENDED
  # expert knobs outside the reference's schema (InternalOptions upstream, cc/kernels/nufft_options.h:92-162):
  # fields of nufft_hip_options by name, e.g. {'tuning': TUNE['ROCFFT'], 'spread_method': 1}; never serialized
  _internal: dict = pydantic.PrivateAttr(default_factory=dict)

  def to_proto(self):
    pb = _proto.OptionsProto()
    pb.debugging = self.debugging.to_proto()
    pb._has_debugging = True   # reference always emits the sub-messages
    pb.fftw = self.fftw.to_proto()
    pb._has_fftw = True
    if self.max_batch_size is not None:
      pb.max_batch_size = int(self.max_batch_size)
    pb.points_range = self.points_range.to_proto()
    return pb

  @classmethod
  def from_proto(cls, pb):
    if isinstance(pb, (bytes, bytearray)):
      pb = _proto.OptionsProto().ParseFromString(bytes(pb))
    obj = cls()
    obj.debugging = DebuggingOptions.from_proto(pb.debugging)
    obj.fftw = FftwOptions.from_proto(pb.fftw)
    obj.max_batch_size = pb.max_batch_size if pb.max_batch_size != 0 else None
    obj.points_range = PointsRange.from_proto(pb.points_range)
    return obj
