// task.hpp -- Qt-free mirror of the reference's Task base class (gui/task.hpp:57-105,
// gui/task.cpp:27-33): same members, Qt signals replaced by std::function observers.
// A Qt adapter derives from QObject and re-emits these as progressUpdate / stageUpdate.
#pragma once

#include <atomic>
#include <functional>
#include <memory>
#include <string>

class Task {
public:
	virtual ~Task() { }

	virtual std::string title() const = 0;
	virtual int numSteps() const = 0;

	// slots
	void run() {                                   // gui/task.cpp:27-33
		cancelled = false;
		if (started) started(this);
		runTask();
		if (finished) finished(this);
	}
	void cancel() { cancelled = true; }
	bool isCancelled() const { return cancelled; }

	// signals
	std::function<void(const Task *)> started, finished;
	std::function<void(int)> progressUpdate;
	std::function<void(const std::string &)> stageUpdate;

protected:
	virtual void runTask() = 0;
	void emitProgress(int step) { if (progressUpdate) progressUpdate(step); }
	void emitStage(const std::string &s) { if (stageUpdate) stageUpdate(s); }
	// the reference's flag is a plain bool written from the GUI thread (gui/task.hpp:104);
	// here it is atomic and is also what the C-ABI polls between kernel launches
	std::atomic<int> cancelled{0};
	const volatile int *cancelFlag() const { return reinterpret_cast<const volatile int *>(&cancelled); }
};

typedef std::shared_ptr<Task> TaskPtr;
