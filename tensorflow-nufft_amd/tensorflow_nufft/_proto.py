"""Minimal proto3 wire codec for the reference's options messages
(tensorflow_nufft/proto/nufft_options.proto:19-32), so that serialized
`options` attrs stay byte-compatible without protoc:

  message FftwOptions      { FftwPlanningRigor planning_rigor = 1; }
  message DebuggingOptions { bool check_points_range = 1; }
  message Options { DebuggingOptions debugging = 1; FftwOptions fftw = 2;
                    int32 max_batch_size = 3; PointsRange points_range = 4; }
"""


def _varint(v):
  v &= (1 << 64) - 1
  out = bytearray()
  while True:
    b = v & 0x7F
    v >>= 7
    if v:
      out.append(b | 0x80)
    else:
      out.append(b)
      return bytes(out)


def _read_varint(buf, pos):
  shift = 0
  val = 0
  while True:
    if pos >= len(buf):
      raise ValueError('truncated varint')
    b = buf[pos]
    pos += 1
    val |= (b & 0x7F) << shift
    if not b & 0x80:
      break
    shift += 7
  if val >= 1 << 63:
    val -= 1 << 64
  return val, pos


def _fields(buf):
  pos = 0
  while pos < len(buf):
    key, pos = _read_varint(buf, pos)
    num, wt = key >> 3, key & 7
    if wt == 0:
      val, pos = _read_varint(buf, pos)
    elif wt == 2:
      n, pos = _read_varint(buf, pos)
      val = bytes(buf[pos:pos + n])
      if len(val) != n:
        raise ValueError('truncated field')
      pos += n
    elif wt == 1:
      val = bytes(buf[pos:pos + 8]); pos += 8
    elif wt == 5:
      val = bytes(buf[pos:pos + 4]); pos += 4
    else:
      raise ValueError(f'unsupported wire type {wt}')
    yield num, wt, val


class _Message:
  def SerializeToString(self):  # pylint: disable=invalid-name
    raise NotImplementedError

  def __eq__(self, other):
    return type(self) is type(other) and self.SerializeToString() == other.SerializeToString()


class FftwOptionsProto(_Message):
  def __init__(self, planning_rigor=0):
    self.planning_rigor = int(planning_rigor)

  def SerializeToString(self):
    return (b'\x08' + _varint(self.planning_rigor)) if self.planning_rigor else b''

  def ParseFromString(self, data):  # pylint: disable=invalid-name
    self.planning_rigor = 0
    for num, wt, val in _fields(data):
      if num == 1 and wt == 0:
        self.planning_rigor = val
    return self


class DebuggingOptionsProto(_Message):
  def __init__(self, check_points_range=False):
    self.check_points_range = bool(check_points_range)

  def SerializeToString(self):
    return b'\x08\x01' if self.check_points_range else b''

  def ParseFromString(self, data):
    self.check_points_range = False
    for num, wt, val in _fields(data):
      if num == 1 and wt == 0:
        self.check_points_range = bool(val)
    return self


class OptionsProto(_Message):
  def __init__(self):
    self.debugging = DebuggingOptionsProto()
    self.fftw = FftwOptionsProto()
    self.max_batch_size = 0
    self.points_range = 0
    self._has_debugging = False
    self._has_fftw = False

  def SerializeToString(self):
    out = b''
    d = self.debugging.SerializeToString()
    if d or self._has_debugging:
      out += b'\x0a' + _varint(len(d)) + d
    f = self.fftw.SerializeToString()
    if f or self._has_fftw:
      out += b'\x12' + _varint(len(f)) + f
    if self.max_batch_size:
      out += b'\x18' + _varint(self.max_batch_size)
    if self.points_range:
      out += b'\x20' + _varint(self.points_range)
    return out

  def ParseFromString(self, data):
    self.__init__()
    for num, wt, val in _fields(data):
      if num == 1 and wt == 2:
        self.debugging.ParseFromString(val); self._has_debugging = True
      elif num == 2 and wt == 2:
        self.fftw.ParseFromString(val); self._has_fftw = True
      elif num == 3 and wt == 0:
        self.max_batch_size = val
      elif num == 4 and wt == 0:
        self.points_range = val
    return self
