"""TEST INFRASTRUCTURE ONLY -- stand-in for the SimPy 4.0.1 subset the reference uses.

The reference (`/root/reference`, requirements.txt:52 pins simpy==4.0.1) drives its
world with SimPy; SimPy is neither vendored nor installable in this image.  This
module re-implements, from the published semantics of SimPy 4.0.1, exactly the
surface the reference touches:

  Environment(), .now, .process(gen), .timeout(delay), .run(until=number|event),
  Event.__and__ / Event.__or__ (Condition all/any), simpy.events.AnyOf (imported
  by runner/check.py only).

Ordering contract reproduced: the queue key is (time, priority, insertion id) with
URGENT=0 < NORMAL=1; Timeout is scheduled NORMAL at now+delay; Process() schedules an
Initialize event URGENT at now; a finished process is scheduled NORMAL at now; a
Condition checks already-processed operands at construction and succeeds NORMAL at
now; run(until=number) schedules its stop event URGENT at that time;
run(until=event) returns at once when the event has already been processed.

It is used only by oracle/refharness/gen_golden.py and tests that pin the C oracle
to the reference *in this container*; nothing in the product imports it and it
never travels as part of the product path.  Because it is a stand-in and not the
real SimPy, golden fixtures are "pinned to reference classes + stand-in" (see
DESIGN.md, section Oracle).
"""
from heapq import heappush, heappop
from itertools import count

URGENT = 0
NORMAL = 1
PENDING = object()


class EmptySchedule(Exception):
    pass


class StopSimulation(Exception):
    @classmethod
    def callback(cls, event):
        if event._ok:
            raise cls(event._value)
        raise event._value


class Event:
    def __init__(self, env):
        self.env = env
        self.callbacks = []
        self._value = PENDING
        self._ok = None

    @property
    def triggered(self):
        return self._value is not PENDING

    @property
    def processed(self):
        return self.callbacks is None

    @property
    def ok(self):
        return self._ok

    @property
    def value(self):
        if self._value is PENDING:
            raise AttributeError("value not yet available")
        return self._value

    def succeed(self, value=None):
        if self._value is not PENDING:
            raise RuntimeError("event already triggered")
        self._ok = True
        self._value = value
        self.env.schedule(self)
        return self

    def fail(self, exception):
        if self._value is not PENDING:
            raise RuntimeError("event already triggered")
        self._ok = False
        self._value = exception
        self.env.schedule(self)
        return self

    def __and__(self, other):
        return Condition(self.env, Condition.all_events, [self, other])

    def __or__(self, other):
        return Condition(self.env, Condition.any_events, [self, other])


class Timeout(Event):
    def __init__(self, env, delay, value=None):
        if delay < 0:
            raise ValueError("negative delay %s" % delay)
        self.env = env
        self.callbacks = []
        self._value = value
        self._delay = delay
        self._ok = True
        env.schedule(self, NORMAL, delay)


class Initialize(Event):
    def __init__(self, env, process):
        self.env = env
        self.callbacks = [process._resume]
        self._value = None
        self._ok = True
        env.schedule(self, URGENT)


class Process(Event):
    def __init__(self, env, generator):
        if not hasattr(generator, "throw"):
            raise ValueError("%s is not a generator" % generator)
        self.env = env
        self.callbacks = []
        self._value = PENDING
        self._ok = None
        self._generator = generator
        self._target = Initialize(env, self)

    @property
    def is_alive(self):
        return self._value is PENDING

    def _resume(self, event):
        self.env._active_proc = self
        while True:
            try:
                if event._ok:
                    event = self._generator.send(event._value)
                else:
                    event._defused = True
                    event = self._generator.throw(event._value)
            except StopIteration as e:
                event = None
                self._ok = True
                self._value = e.args[0] if len(e.args) else None
                self.env.schedule(self)
                break
            except BaseException as e:
                event = None
                self._ok = False
                self._value = e
                self.env.schedule(self)
                break
            try:
                if event.callbacks is not None:
                    event.callbacks.append(self._resume)
                    break
            except AttributeError:
                raise RuntimeError("invalid yield value %r" % (event,))
            # event already processed: feed its value straight back in
        self._target = event
        self.env._active_proc = None


class ConditionValue:
    def __init__(self):
        self.events = []


class Condition(Event):
    def __init__(self, env, evaluate, events):
        super().__init__(env)
        self._evaluate = evaluate
        self._events = tuple(events)
        self._count = 0
        if not self._events:
            self.succeed(ConditionValue())
            return
        for event in self._events:
            if event.callbacks is None:
                self._check(event)
            else:
                event.callbacks.append(self._check)
        self.callbacks.append(self._build_value)

    def _build_value(self, event):
        if event._ok:
            self._value = ConditionValue()

    def _check(self, event):
        if self._value is not PENDING:
            return
        self._count += 1
        if not event._ok:
            event._defused = True
            self.fail(event._value)
        elif self._evaluate(self._events, self._count):
            self.succeed()

    @staticmethod
    def all_events(events, count):
        return len(events) == count

    @staticmethod
    def any_events(events, count):
        return count > 0 or len(events) == 0


class AllOf(Condition):
    def __init__(self, env, events):
        super().__init__(env, Condition.all_events, events)


class AnyOf(Condition):
    def __init__(self, env, events):
        super().__init__(env, Condition.any_events, events)


class Environment:
    def __init__(self, initial_time=0):
        self._now = initial_time
        self._queue = []
        self._eid = count()
        self._active_proc = None
        self.n_events = 0

    @property
    def now(self):
        return self._now

    def schedule(self, event, priority=NORMAL, delay=0):
        heappush(self._queue, (self._now + delay, priority, next(self._eid), event))

    def process(self, generator):
        return Process(self, generator)

    def timeout(self, delay=0, value=None):
        return Timeout(self, delay, value)

    def event(self):
        return Event(self)

    def all_of(self, events):
        return AllOf(self, events)

    def any_of(self, events):
        return AnyOf(self, events)

    def peek(self):
        return self._queue[0][0] if self._queue else float("inf")

    def step(self):
        try:
            self._now, _, _, event = heappop(self._queue)
        except IndexError:
            raise EmptySchedule()
        self.n_events += 1
        callbacks, event.callbacks = event.callbacks, None
        for callback in callbacks:
            callback(event)
        if not event._ok and not hasattr(event, "_defused"):
            raise event._value

    def run(self, until=None):
        if until is not None:
            if not isinstance(until, Event):
                at = float(until)
                if at <= self.now:
                    raise ValueError("until(=%s) must be > the current simulation time" % at)
                until = Event(self)
                until._ok = True
                until._value = None
                self.schedule(until, URGENT, at - self.now)
            elif until.callbacks is None:
                return until.value
            until.callbacks.append(StopSimulation.callback)
        try:
            while True:
                self.step()
        except StopSimulation as exc:
            return exc.args[0]
        except EmptySchedule:
            if until is not None and not until.triggered:
                raise RuntimeError("no scheduled events left but until event was not triggered")
        return None
