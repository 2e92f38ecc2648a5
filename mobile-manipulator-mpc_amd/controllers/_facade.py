"""Tiny stand-ins for the CasADi objects the reference's driver reaches into
(interface_wholebody_qref.py:166-167: ``controller.opti.subject_to(controller.X[N,:2] ==
controller.X_ref[N,:2])``).  Only that one expression is understood."""


class _Slice:
    def __init__(self, name, key):
        self.name, self.key = name, key

    def __eq__(self, other):
        return _Eq(self, other)

    __hash__ = None


class _Eq:
    def __init__(self, lhs, rhs):
        self.lhs, self.rhs = lhs, rhs


class Sym:
    def __init__(self, name):
        self.name = name

    def __getitem__(self, key):
        return _Slice(self.name, key)


class OptiFacade:
    def __init__(self, controller):
        self._c = controller

    def subject_to(self, expr):
        N = self._c.N
        ok = (isinstance(expr, _Eq) and isinstance(expr.lhs, _Slice) and isinstance(expr.rhs, _Slice)
              and expr.lhs.name == "X" and expr.rhs.name == "X_ref"
              and expr.lhs.key == (N, slice(None, 2, None)) and expr.rhs.key == (N, slice(None, 2, None)))
        if not ok:
            raise NotImplementedError("only X[N,:2] == X_ref[N,:2] (interface_wholebody_qref.py:166-167) is supported")
        self._c._set_terminal_xy_equality(True)
