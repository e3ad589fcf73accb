"""Build-container-only stand-in for torchdiffeq==0.2.3 (forward pass, fixed-grid solvers).

Used only by tools/gen_golden.py to import the reference; see ../README.md.
euler:  y1 = y0 + dt*f(t0,y0)
rk4  :  torchdiffeq 0.2.3 `rk4_alt_step_func` (3/8 rule):
        k1=f(t0,y); k2=f(t0+dt/3, y+dt*k1/3); k3=f(t0+2dt/3, y+dt*(k2-k1/3));
        k4=f(t1, y+dt*(k1-k2+k3)); dy=(k1+3*(k2+k3)+k4)*dt*0.125
"""
import torch

_one_third = 1.0 / 3.0
_two_thirds = 2.0 / 3.0


def _euler(f, t0, dt, t1, y):
    return dt * f(t0, y)


def _rk4(f, t0, dt, t1, y):
    k1 = f(t0, y)
    k2 = f(t0 + dt * _one_third, y + dt * k1 * _one_third)
    k3 = f(t0 + dt * _two_thirds, y + dt * (k2 - k1 * _one_third))
    k4 = f(t1, y + dt * (k1 - k2 + k3))
    return (k1 + 3 * (k2 + k3) + k4) * dt * 0.125


def odeint(func, y0, t, method='dopri5', **kw):
    step = {'euler': _euler, 'rk4': _rk4}[method]
    sol = [y0]
    y = y0
    with torch.no_grad():
        for t0, t1 in zip(t[:-1], t[1:]):
            y = y + step(func, t0, t1 - t0, t1, y)
            sol.append(y)
    return torch.stack(sol)


odeint_adjoint = odeint
