"""Rescue-Prime workload generator for the STARK caller tests (reference code/rescue_prime.py:5-273 semantics).

Out of scope for the GPU path: it only produces the 28-row trace, the AIR and the boundary
constraints that drive FastStark in tests.  All parameters (MDS matrices, the 108 round constants, known-answer
hashes) are DATA read from tests/golden/rescue_prime_params.json, which make_golden.py wrote from the reference.
"""
import json
import os

from algebra import Field, FieldElement
from univariate import Polynomial
from multivariate import MPolynomial

_PARAMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rescue_prime_params.json")


class RescuePrime:
    def __init__(self):
        with open(_PARAMS) as f:
            prm = json.load(f)
        self.p = int(prm["p"])
        self.field = Field(self.p)
        self.m, self.N, self.alpha, self.alphainv = prm["m"], prm["N"], prm["alpha"], int(prm["alphainv"])
        self.rate = self.capacity = 1
        fe = lambda v: FieldElement(int(v), self.field)
        self.MDS = [[fe(v) for v in row] for row in prm["MDS"]]
        self.MDSinv = [[fe(v) for v in row] for row in prm["MDSinv"]]
        self.round_constants = [fe(v) for v in prm["round_constants"]]
        self.kat_hash = [(int(a), int(b)) for a, b in prm["kat_hash"]]

    def _half_round(self, state, exponent, constants_offset):
        powered = [s ^ exponent for s in state]
        mixed = []
        for i in range(self.m):
            acc = self.field.zero()
            for j in range(self.m):
                acc = acc + self.MDS[i][j] * powered[j]
            mixed.append(acc + self.round_constants[constants_offset + i])
        return mixed

    def _states(self, input_element):
        state = [input_element] + [self.field.zero()] * (self.m - 1)
        yield state
        for r in range(self.N):
            state = self._half_round(state, self.alpha, 2 * r * self.m)
            state = self._half_round(state, self.alphainv, 2 * r * self.m + self.m)
            yield state

    def hash(self, input_element):
        last = None
        for last in self._states(input_element):
            pass
        return last[0]

    def trace(self, input_element):
        return [[s for s in state] for state in self._states(input_element)]

    def boundary_constraints(self, output_element):
        # capacity starts at zero; the rate part ends at the claimed output
        return [(0, 1, self.field.zero()), (self.N, 0, output_element)]

    def round_constants_polynomials(self, omicron):
        domain = [omicron ^ r for r in range(self.N)]

        def lifted(offset):
            polys = []
            for i in range(self.m):
                values = [self.round_constants[2 * r * self.m + offset + i] for r in range(self.N)]
                polys.append(MPolynomial.lift(Polynomial.interpolate_domain(domain, values), 0))
            return polys

        return lifted(0), lifted(self.m)

    def transition_constraints(self, omicron):
        first_step_constants, second_step_constants = self.round_constants_polynomials(omicron)
        variables = MPolynomial.variables(1 + 2 * self.m, self.field)
        previous_state = variables[1:(1 + self.m)]
        next_state = variables[(1 + self.m):(1 + 2 * self.m)]
        air = []
        for i in range(self.m):
            lhs = MPolynomial.constant(self.field.zero())
            for k in range(self.m):
                lhs = lhs + MPolynomial.constant(self.MDS[i][k]) * (previous_state[k] ^ self.alpha)
            lhs = lhs + first_step_constants[i]
            rhs = MPolynomial.constant(self.field.zero())
            for k in range(self.m):
                rhs = rhs + MPolynomial.constant(self.MDSinv[i][k]) * (next_state[k] - second_step_constants[k])
            rhs = rhs ^ self.alpha
            air.append(lhs - rhs)
        return air
