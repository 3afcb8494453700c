"""Second, independent restatement of the reference ASSEMBLY in numpy / pure Python.

TEST INFRASTRUCTURE ONLY.  Small cases only (pure-Python loops).  It exists so that the C++
oracle (fi_oracle.cpp) is not its own only witness: tests/test_oracle_cross.py requires the two to
produce identical (row, col, fp32 value) triplets and fp32 rhs on a matrix of inputs.

Written from the reference semantics, one function per reference function:
  add_equation                          sparse_linear.cpp:34-50
  multilerp                             field_interpolation.cpp:15-55
  add_value_constraint                  field_interpolation.cpp:57-80
  add_value_constraint_nearest_neighbor field_interpolation.cpp:82-107
  cell_index                            field_interpolation.cpp:110-121
  add_gradient_constraint               field_interpolation.cpp:123-240
  add_model_constraint / add_field_constraints  field_interpolation.cpp:243-341
  add_points / sdf_from_points          field_interpolation.cpp:343-400
All arithmetic on coefficients is done in numpy float32 scalars so that products and sums round
exactly as the reference's `float` expressions do.
"""
import itertools
import math

import numpy as np

F = np.float32


class PyField:
    def __init__(self, sizes):
        self.sizes = [int(s) for s in sizes]
        self.strides = []
        s = 1
        for n in self.sizes:
            self.strides.append(s)
            s *= n
        self.trip = []   # (row, col, float32)
        self.rhs = []    # float32

    @property
    def D(self):
        return len(self.sizes)

    # sparse_linear.cpp:34-50
    def add_equation(self, weight, rhs, pairs):
        weight, rhs = F(weight), F(rhs)
        if weight == 0:
            return
        row = len(self.rhs)
        kept = False
        for col, v in pairs:
            v = F(v)
            if v != 0:
                self.trip.append((row, int(col), F(v * weight)))
                kept = True
        if kept:
            self.rhs.append(F(rhs * weight))

    # field_interpolation.cpp:15-55
    def multilerp(self, pos, extra):
        pos = [F(p) for p in pos]
        fl = [int(math.floor(float(p))) for p in pos]
        t = [F(p - F(f)) for p, f in zip(pos, fl)]
        out = []
        for bits in itertools.product((0, 1), repeat=self.D):
            bits = bits[::-1]  # product varies the LAST element fastest; corner bit 0 <-> axis 0
            idx, w, ok = 0, F(1), True
            for d in range(self.D):
                q = fl[d] + bits[d]
                idx += self.strides[d] * q
                w = F(w * (t[d] if bits[d] else F(F(1) - t[d])))
                ok = ok and 0 <= q and q + extra < self.sizes[d]
            if ok:
                out.append((idx, w))
        return out

    # field_interpolation.cpp:57-80
    def add_value_constraint(self, pos, value, cw):
        cw, value = F(cw), F(value)
        if cw == 0:
            return False
        samples = self.multilerp(pos, 0)
        if not samples:
            return False
        row = len(self.rhs)
        total = F(0)
        for idx, w in samples:
            c = F(w * cw)
            self.trip.append((row, idx, c))
            total = F(total + c)
        self.rhs.append(F(total * value))
        return True

    # field_interpolation.cpp:82-107
    def add_value_constraint_nearest_neighbor(self, pos, grad, value, cw):
        idx, along = 0, F(0)
        for d in range(self.D):
            p = F(pos[d])
            q = int(math.floor(abs(float(p)) + 0.5) * (1 if p >= 0 else -1))  # std::round
            if q < 0 or self.sizes[d] <= q:
                return False
            along = F(along + F(F(p - F(q)) * F(grad[d])))
            idx += q * self.strides[d]
        self.add_equation(cw, F(F(value) - along), [(idx, 1.0)])
        return True

    # field_interpolation.cpp:110-121
    def cell_index(self, pos):
        idx = 0
        for d in range(self.D):
            q = int(math.floor(float(F(pos[d]))))
            if not (0 <= q and q + 1 < self.sizes[d]):
                return -1
            idx += q * self.strides[d]
        return idx

    # field_interpolation.cpp:123-240
    def add_gradient_constraint(self, pos, grad, cw, kernel):
        cw = F(cw)
        if cw == 0:
            return False
        D = self.D
        if kernel == 0:
            o = self.cell_index(pos)
            if o < 0:
                return False
            for d in range(D):
                self.add_equation(cw, grad[d], [(o, -1.0), (o + self.strides[d], +1.0)])
            return True
        if kernel == 1:
            o = self.cell_index(pos)
            if o < 0:
                return False
            nc = 1 << D
            for d in range(D):
                row = len(self.rhs)
                term = F(F(cw * F(2)) / F(nc))
                for c in range(nc):
                    col = o + sum(self.strides[a] * ((c >> a) & 1) for a in range(D))
                    sign = F(1) if (c >> d) & 1 else F(-1)
                    self.trip.append((row, col, F(sign * term)))
                self.rhs.append(F(cw * F(grad[d])))
            return True
        if kernel == 2:
            shifted = [F(F(p) - F(0.5)) for p in pos]
            samples = self.multilerp(shifted, 1)
            if not samples:
                return False
            for d in range(D):
                row = len(self.rhs)
                total = F(0)
                for idx, w in samples:
                    c = F(w * cw)
                    self.trip.append((row, idx, F(-c)))
                    self.trip.append((row, idx + self.strides[d], c))
                    total = F(total + c)
                self.rhs.append(F(total * F(grad[d])))
            return True
        raise ValueError("Unknown gradient kernel")

    # field_interpolation.cpp:243-341
    def add_field_constraints(self, w):
        N = int(np.prod(self.sizes))
        pascal = {1: (w.model_1, [-1, 1]), 2: (w.model_2, [1, -2, 1]),
                  3: (w.model_3, [1, -3, 3, -1]), 4: (w.model_4, [1, -4, 6, -4, 1])}
        for index in range(N):
            coord, rest = [], index
            for n in self.sizes:
                coord.append(rest % n)
                rest //= n
            for d in range(self.D):
                size, st, c = self.sizes[d], self.strides[d], coord[d]
                if w.model_0 > 0 and 0 <= c < size:
                    self.add_equation(w.model_0, 0.0, [(index, 1.0)])
                for order in (1, 2, 3, 4):
                    wt, coefs = pascal[order]
                    if wt > 0 and 0 <= c and c + order < size:
                        self.add_equation(wt, 0.0, [(index + k * st, float(v)) for k, v in enumerate(coefs)])
                if w.gradient_smoothness > 0 and 0 <= c and c + 1 < size:
                    for o in range(self.D):
                        if o == d or coord[o] + 1 >= self.sizes[o]:
                            continue
                        so = self.strides[o]
                        self.add_equation(w.gradient_smoothness, 0.0,
                                          [(index, -1.0), (index + st, 1.0), (index + so, 1.0),
                                           (index + so + st, -1.0)])

    # field_interpolation.cpp:343-371
    def add_points(self, vw, vk, gw, gk, positions, normals=None, point_weights=None):
        pos = np.asarray(positions, np.float32).reshape(-1, self.D)
        nrm = None if normals is None else np.asarray(normals, np.float32).reshape(-1, self.D)
        for i in range(len(pos)):
            w = F(1) if point_weights is None else F(point_weights[i])
            if vk == 0:
                if nrm is None:
                    raise ValueError("normals required")
                self.add_value_constraint_nearest_neighbor(pos[i], nrm[i], 0.0, F(w * F(vw)))
            else:
                self.add_value_constraint(pos[i], 0.0, F(w * F(vw)))
            if nrm is not None:
                self.add_gradient_constraint(pos[i], nrm[i], F(w * F(gw)), gk)

    def arrays(self):
        rows = np.array([t[0] for t in self.trip], np.int32)
        cols = np.array([t[1] for t in self.trip], np.int32)
        vals = np.array([t[2] for t in self.trip], np.float32)
        return rows, cols, vals, np.array(self.rhs, np.float32)

    def dense(self):
        """(A, b) in float64 with duplicates summed."""
        N = int(np.prod(self.sizes))
        A = np.zeros((len(self.rhs), N))
        for r, c, v in self.trip:
            A[r, c] += float(v)
        return A, np.array(self.rhs, np.float64)


# field_interpolation.cpp:373-400
def sdf_from_points(sizes, weights, positions, normals=None, point_weights=None):
    f = PyField(sizes)
    f.add_field_constraints(weights)
    f.add_points(weights.data_pos, weights.value_kernel, weights.data_gradient, weights.gradient_kernel,
                 positions, normals, point_weights)
    return f
